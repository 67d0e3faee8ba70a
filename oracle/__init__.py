"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/bn254.py header).  Not part of the product."""
