"""GPU: the multi-rank branches of the library on ONE GPU, over tests/cpp/rccl_stub.cpp — TEST INFRASTRUCTURE standing in for
RCCL (the real one refuses two ranks on one device).  The file's name makes it collect LAST: these tests depend on a stand-in,
and under `pytest -x` a failure here must cost only itself, never the parity tests (one such test hid 98 others).

What runs through the stand-in, all inside the C ABI:
  * h2agg_comm_create + h2agg_allgather_add_points with nctx == world (ncclCommInitAll, ncclGroupStart / ncclAllGather per
    context / ncclGroupEnd; csrc/comm.inc) — a C++ driver, the stand-in found under its soname librccl.so.1;
  * h2agg_verify_aggregation_sharded with shard->allgather = NULL (h2agg_comm_init_rank + both exchanges as ncclAllGather on
    the context's stream; csrc/verifier.inc shard_allgather) with the ranks as THREADS and as PROCESSES (hipIpc), named to the
    library with H2AGG_RCCL_LIB — verify.rs:909-913, :924 (lambda) and :926-938 (the fold), SURVEY.md 8(e);
  * bench.py --gpus 2 with both ranks on device 0: `aggregate` and `from_bytes_sharded` over the library's communicator of TWO
    ranks in two processes (`rccl_ranks: 2`), equal to the one-rank recomputation.
The stand-in's all-gather is stream-ordered like RCCL's (see its header), so a missing ordering in the library shows."""
import json
import os
import shutil
import socket
import subprocess
import sys

import pytest

import __graft_entry__ as entry

ROOT, PKG = entry.ROOT, entry.PKG_DIR


def build_stub(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    stub = str(tmp_path / "librccl.so.1")
    subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-Wl,-soname,librccl.so.1", os.path.join(ROOT, "tests", "cpp", "rccl_stub.cpp"), "-o", stub, "-lrt"],
                   check=True, capture_output=True, text=True)
    return stub


def _build(tmp_path):
    entry.build()
    stub = build_stub(tmp_path)
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


def run_ranks(stub, world, mode, how, loops=1, timeout=900):
    env = {k: v for k, v in os.environ.items() if k != "H2AGG_RCCL_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_stub_ranks.py"), stub, str(world), mode, how, str(loops)],
                       capture_output=True, text=True, timeout=timeout, env=env)
    return r.returncode == 0 and "RCCL-STUB-RANKS-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["threads", "procs"])
@pytest.mark.parametrize("world,mode", [(2, "ok"), (3, "ok"), (2, "fail1"), (3, "fail2"), (2, "fail3"), (3, "fail4")])
def test_sharded_over_the_librarys_own_transport(tmp_path, world, mode, how):
    """shard->allgather = NULL at world > 1 on one GPU (see tests/rccl_stub_ranks.py): every rank returns the one-context
    call's pair, lambda and verdict; a rank that fails — before, between or INSIDE the exchanges — hangs nobody"""
    ok, log = run_ranks(build_stub(tmp_path), world, mode, how, loops=3 if mode == "ok" else 1)
    assert ok, log


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


BENCH = ["--log2n", "16", "--steps", "2", "--warmup", "1", "--spinup", "0", "--no-cpu-baseline", "--no-pcie-leg", "--agg-instance-log2", "12",
         "--agg-config4", "0"]


def run_bench(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_processes_over_the_librarys_communicator(tmp_path):
    """`python bench.py --gpus 2` on one GPU: two processes (torch.distributed over gloo for the launch), the library's own
    communicator of TWO ranks (h2agg_comm_init_rank in each process, the stand-in's process mode) under both aggregation legs —
    the rehearsal of what the driver's 8-GPU run does first."""
    stub = build_stub(tmp_path)
    env = dict(os.environ, H2AGG_DIST_BACKEND="gloo", H2AGG_RCCL_LIB=stub, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    two = run_bench([sys.executable, "bench.py", "--gpus", "2", "--agg-proofs", "2"] + BENCH, env)
    plain = {k: v for k, v in os.environ.items() if k not in ("H2AGG_RCCL_LIB", "H2AGG_DIST_BACKEND")}
    one = run_bench([sys.executable, "bench.py", "--gpus", "1", "--agg-proofs", "4"] + BENCH, plain)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    a2, a1 = two["aggregate"], one["aggregate"]
    assert "error" not in a2 and "error" not in a1, (a2, a1)
    assert a2["rccl_ranks"] == 2 and "h2agg_allgather_add_points" in a2["exchange"] and a2["rccl_library"] == stub
    assert a2["proofs"] == a1["proofs"] == 4 and a2["final_pair_sha"] == a1["final_pair_sha"]
    assert a2["config3"]["rccl_ranks"] == 2 and a2["config3"]["one_rank_recomputation"]
    f2, f1 = a2["from_bytes_sharded"], a1["from_bytes_sharded"]
    assert "error" not in f2 and "error" not in f1, (f2, f1)
    assert f2["rccl_ranks"] == 2 and f1["rccl_ranks"] == 1 and f2["rccl_library"] == stub
    assert "equals_one_context_call" in f2 and f2["proofs"] == f1["proofs"] == 4
    assert f2["final_pair_sha"] == f1["final_pair_sha"] and f2["lambda_sha"] == f1["lambda_sha"]
