#!/usr/bin/env python3
"""Soak of `python bench.py --gpus 2` on ONE GPU over the library's own communicator: two processes, gloo for
the launch, tests/cpp/rccl_stub.cpp in its process mode under h2agg_comm_init_rank — N consecutive runs, each of which must
print one line with n_gpus = 2, rccl_ranks = 2 on both aggregation legs and the final pairs / lambda of the one-rank run of
the same four proofs.      python tools/soak_bench_ranks.py [N] [out.txt]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "soak_bench_ranks.txt")
os.makedirs("/tmp/soakb", exist_ok=True)
stub = "/tmp/soakb/librccl_standin.so"
subprocess.run(["hipcc", "-O1", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "rccl_stub.cpp"), "-o", stub, "-lrt"], check=True)
COMMON = ["--log2n", "16", "--steps", "2", "--warmup", "1", "--spinup", "0", "--no-cpu-baseline", "--no-pcie-leg", "--agg-instance-log2", "12",
          "--agg-config4", "0"]


def run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        return None, (p.stderr[-1500:] + p.stdout[-500:])
    return json.loads(lines[0]), ""


plain = {k: v for k, v in os.environ.items() if k not in ("H2AGG_RCCL_LIB", "H2AGG_DIST_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
one, err = run([sys.executable, "bench.py", "--gpus", "1", "--agg-proofs", "4"] + COMMON, plain)
assert one, err
a1, f1 = one["aggregate"], one["aggregate"]["from_bytes_sharded"]
env = dict(plain, H2AGG_DIST_BACKEND="gloo", H2AGG_RCCL_LIB=stub, MASTER_ADDR="127.0.0.1")
ok = bad = 0
t0 = time.time()
with open(out_path, "w") as f:
    for i in range(N):
        two, err = run([sys.executable, "bench.py", "--gpus", "2", "--agg-proofs", "2"] + COMMON, env)
        good = False
        if two:
            a2 = two["aggregate"]
            f2 = a2.get("from_bytes_sharded", {})
            good = (two["n_gpus"] == 2 and "error" not in a2 and "error" not in f2 and a2["rccl_ranks"] == 2 and f2.get("rccl_ranks") == 2
                    and a2["final_pair_sha"] == a1["final_pair_sha"] and f2["final_pair_sha"] == f1["final_pair_sha"]
                    and f2["lambda_sha"] == f1["lambda_sha"] and bool(a2["config3"]["one_rank_recomputation"]))
        ok, bad = ok + good, bad + (not good)
        if not good:
            f.write("--- run %d FAILED: %s\n" % (i, err or json.dumps(two.get("aggregate", {}))[:1500]))
            f.flush()
    line = ("bench.py --gpus 2 on one GPU, two processes, the library's communicator over the stand-in (rccl_ranks 2 on `aggregate` and "
            "`from_bytes_sharded`, pairs and lambda equal to the one-rank run): %d passed, %d failed of %d consecutive runs (%d s)"
            % (ok, bad, N, time.time() - t0))
    f.write(line + "\n")
print(line)
sys.exit(1 if bad else 0)
