"""Offline differential run of the schema path (random ASTs, tests/test_gpu_schema.py::test_random_schema_trees_vs_oracle)
over many seeds; not collected by pytest.  795 seeds clean in 150 s on the GPU box."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
import tests.test_gpu_schema as T
pkg = entry.load_package(); eng = pkg.H2Agg(0)
t0 = time.time(); n = 0
for seed in range(12, 100000):
    if time.time() - t0 > 150: break
    T.test_random_schema_trees_vs_oracle(eng, pkg, seed); n += 1
print("schema fuzz ok:", n, "seeds")
