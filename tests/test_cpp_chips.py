"""The C++ host side above the C ABI (include/h2agg_chips.hpp: the reference's trait trio as header-only classes).
CPU: the header and the driver compile and link against libh2agg.so.  GPU: the driver runs every trait method against
the C oracle (oracle/liboracle_bn254.so, loaded by the driver with dlopen — test infrastructure)."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as entry

ROOT = entry.ROOT
PKG = entry.PKG_DIR


def _build(tmp_path):
    entry.build()
    from oracle import cref
    cref.build()
    exe = str(tmp_path / "chips_driver")
    cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "chips_driver.cpp"), "-o", exe, "-L", PKG, "-lh2agg", "-ldl",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_chip_classes_compile_and_link(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    # without the oracle path the driver stops before it touches the device
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_cpp_chip_classes_match_the_oracle(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "oracle", "liboracle_bn254.so")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "chips ok" in r.stdout, r.stdout + r.stderr
