"""GPU: the ONE-PROCESS, N-CONTEXT branch of the exchange (h2agg_comm_create + h2agg_allgather_add_points with nctx == world:
ncclCommInitAll, ncclGroupStart / one ncclAllGather per context / ncclGroupEnd; csrc/comm.inc) run on a one-GPU box: a C++
driver links libh2agg.so and loads tests/cpp/rccl_stub.cpp — test infrastructure with the soname librccl.so.1 whose
"ranks" all live on device 0 — before the library resolves RCCL.  (The real RCCL refuses two ranks on one device; with real
devices the same branch runs in tests/test_gpu_comm.py::test_single_process_group_two_devices.)"""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as entry

ROOT, PKG = entry.ROOT, entry.PKG_DIR


def _build(tmp_path):
    entry.build()
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    stub = str(tmp_path / "librccl.so.1")
    subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-Wl,-soname,librccl.so.1", os.path.join(ROOT, "tests", "cpp", "rccl_stub.cpp"), "-o", stub],
                   check=True, capture_output=True, text=True)
    exe = str(tmp_path / "comm_group_driver")
    subprocess.run([shutil.which("g++") or "g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "comm_group_driver.cpp"), "-o", exe, "-L", PKG, "-lh2agg", "-ldl",
                    "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    return exe, stub


def test_comm_group_driver_and_stub_build(tmp_path):
    exe, stub = _build(tmp_path)
    assert os.path.exists(exe) and os.path.exists(stub)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_single_process_group_branch_on_one_device(tmp_path, world):
    exe, stub = _build(tmp_path)
    r = subprocess.run([exe, stub, str(world)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "comm group ok" in r.stdout, r.stdout + r.stderr
