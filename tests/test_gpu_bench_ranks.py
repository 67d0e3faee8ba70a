"""GPU: bench.py's multi-rank control flow on ONE GPU.  The driver's 8-GPU runs go over RCCL; here the
same launch line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) runs with
H2AGG_DIST_BACKEND=gloo (collectives through host tensors, both ranks on device 0 — bench.py documents the mode) and must
  * print ONE well-formed JSON line from rank 0 with n_gpus = 2,
  * say which exchange the aggregate leg took (`aggregate.exchange`),
  * fold to the SAME final pair as one rank holding all the proofs (2 ranks x 2 proofs == 1 rank x 4 proofs), bit for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--log2n", "16", "--steps", "2", "--warmup", "1", "--spinup", "0", "--no-cpu-baseline", "--no-pcie-leg",
          "--agg-instance-log2", "12"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_reproduce_one_rank():
    env = dict(os.environ, H2AGG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--agg-proofs", "2"] + COMMON, env)
    one = run([sys.executable, "bench.py", "--gpus", "1", "--agg-proofs", "4"] + COMMON, dict(os.environ))
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["value"] > 0 and two["steps"] == 2
    a2, a1 = two["aggregate"], one["aggregate"]
    assert "error" not in a2 and "error" not in a1, (a2, a1)
    assert a2["proofs"] == a1["proofs"] == 4
    assert a2["final_pair_sha"] == a1["final_pair_sha"]
    assert "all_gather" in a2["exchange"] or "allgather" in a2["exchange"]
    assert a2["rccl_ranks"] == 0            # gloo run: the torch path, and the line says so
    assert a1["exchange"].startswith("none")
    # the aggregation from PROOF BYTES, sharded: h2agg_verify_aggregation_sharded with both exchanges inside the C ABI.  Two
    # ranks x 2 proofs and one rank x 4 proofs are the same four proofs: the same pair and the same lambda.
    f2, f1 = a2["from_bytes_sharded"], a1["from_bytes_sharded"]
    assert f2["proofs"] == f1["proofs"] == 4
    assert f2["final_pair_sha"] == f1["final_pair_sha"] and f2["lambda_sha"] == f1["lambda_sha"]
    assert f1["rccl_ranks"] == 1            # one GPU: a one-rank RCCL communicator under shard->allgather = NULL
    assert f2["rccl_ranks"] == 0 and "torch.distributed" in f2["transport"]
    for leg in (a1, a2, f1, f2):
        assert leg["latency"]["repetitions"] >= 32 and leg["latency"]["p95_s"] >= leg["latency"]["p50_s"]


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` the way the driver's N = 1 command is shaped (no torch.distributed.run, WORLD_SIZE unset) used
    to run ONE rank and print n_gpus = 1: it now re-launches itself with one rank per GPU and prints n_gpus = 2 —
    the same aggregation as the launcher-started run."""
    env = dict(os.environ, H2AGG_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    two = run([sys.executable, "bench.py", "--gpus", "2", "--agg-proofs", "2"] + COMMON, env)
    assert two["n_gpus"] == 2
    one = run([sys.executable, "bench.py", "--agg-proofs", "4"] + COMMON, dict(os.environ))     # no --gpus at all: one rank
    assert one["n_gpus"] == 1
    assert two["aggregate"]["final_pair_sha"] == one["aggregate"]["final_pair_sha"]
    assert two["aggregate"]["from_bytes_sharded"]["final_pair_sha"] == one["aggregate"]["from_bytes_sharded"]["final_pair_sha"]


def test_a_launch_that_does_not_match_gpus_is_refused():
    env = dict(os.environ, H2AGG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), "bench.py", "--gpus", "4", "--agg-proofs", "0"] + COMMON,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
