"""Child process of tests/test_gpu_zz_standins.py (not collected by pytest).

    python tests/rccl_stub_ranks.py <RCCL stand-in .so> <world> <ok | fail1 | fail2 | fail3 | fail4> [threads | procs] [loops]

h2agg_verify_aggregation_sharded with shard->allgather = NULL — the library's OWN transport: h2agg_comm_init_rank, then both
exchanges as ncclAllGather on the context's stream between a host-to-device and a device-to-host copy (csrc/verifier.inc
shard_allgather) — at world > 1 on a one-GPU box.  RCCL is tests/cpp/rccl_stub.cpp (stream-ordered all-gather; see its
header), named to the library with H2AGG_RCCL_LIB; this process never imports torch.

  threads   the ranks are threads of this process, one context each
  procs     the ranks are PROCESSES sharing the device (what one process per GPU does): this process computes the
            one-context answer and the unique id, starts `world` copies of itself (`--rank r --uid hex`) and compares

Every rank must return the one-context call's pair, lambda and verdict; with an injected failure on rank 0 (debug key
shard_fail: 1 before exchange 1, 2 between the exchanges, 3 / 4 inside exchange 1 / 2 before its all-gather) every rank must
return — rank 0 its own error, the others H2AGG_ERR_PEER.  `loops` repeats the sharded call in the same contexts (soak)."""
import ctypes
import importlib
import os
import pickle
import subprocess
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--"))
stub, world, mode = argv[0], int(argv[1]), argv[2]
how = argv[3] if len(argv) > 3 else "threads"
loops = int(argv[4]) if len(argv) > 4 else 1
os.environ["H2AGG_RCCL_LIB"] = stub
_keep = ctypes.CDLL(stub)

import __graft_entry__ as entry  # noqa: E402
from tests.test_gpu_sharded import product_args  # noqa: E402
from tests.test_gpu_verifier import run_product  # noqa: E402
from tests.test_pairing_capi import g2b  # noqa: E402
from tests.test_sharded_from_bytes import shard_batch  # noqa: E402
from tests.test_verifier_pipeline import SHAPES, make_batch  # noqa: E402

assert "torch" not in sys.modules
pkg = entry.load_package()
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
setup, circuits = make_batch(0x7C0 + world, [SHAPES[0]], 2 * world - 1)      # odd: the last rank holds one proof less
FAIL = {"fail1": 1, "fail2": 2, "fail3": 3, "fail4": 4}


def rank_main(rank, uid, out):
    eng = pkg.H2Agg(0)
    try:
        eng.comm_init_rank(uid, rank, world)
        assert eng.comm_size() == world and eng.comm_rank() == rank
        if mode != "ok" and rank == 0:
            eng.debug_configure("shard_fail", FAIL[mode])
        local, gidx, n_total = shard_batch(circuits, world, rank)
        table, vks, arg = product_args(ver, eng, setup, local)
        try:
            for _ in range(loops):
                got = ver.verify_aggregation_sharded(eng, arg, gidx, n_total, rank, world, None, g2b(setup.s_g2), g2b(setup.g2))
                if out[rank] is not None and got != out[rank]:
                    raise AssertionError("repetitions of rank %d disagree" % rank)
                out[rank] = got
        finally:
            for vk in vks:
                vk.close()
            eng.bases_free(table)
    except BaseException as e:   # noqa
        out[rank] = e
    finally:
        eng.close()


if "rank" in opts:          # a rank of the procs mode: one result on stdout, for the parent
    out = [None] * world
    rank_main(int(opts["rank"]), bytes.fromhex(opts["uid"]), out)
    res = out[int(opts["rank"])]
    if isinstance(res, BaseException):
        res = ("ERR", getattr(res, "code", None), repr(res))
    print("RANK-RESULT " + pickle.dumps((res, _keep.rccl_stub_allgathers())).hex(), flush=True)
    os._exit(0)

eng0 = pkg.H2Agg(0)
want = run_product(pkg, eng0, setup, circuits)
uid = pkg.H2Agg.comm_unique_id()
out = [None] * world
if how == "threads":
    ts = [threading.Thread(target=rank_main, args=(r, uid, out), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    if any(t.is_alive() for t in ts):
        print("HANG: a rank is still inside an exchange", flush=True)
        os._exit(3)
    n_allgathers = _keep.rccl_stub_allgathers()
    codes = [getattr(o, "code", None) for o in out]
else:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), stub, str(world), mode, "procs", str(loops), "--rank=%d" % r,
                            "--uid=" + uid.hex()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    n_allgathers, codes = 0, [None] * world
    for r, p in enumerate(ps):
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            print("HANG: rank %d is still inside an exchange" % r, flush=True)
            os._exit(3)
        lines = [ln for ln in so.splitlines() if ln.startswith("RANK-RESULT ")]
        if not lines:
            out[r] = "rank %d died: %s" % (r, se[-1500:])
            continue
        res, n_ag = pickle.loads(bytes.fromhex(lines[0].split()[1]))
        n_allgathers += n_ag
        if isinstance(res, tuple) and res and res[0] == "ERR":
            codes[r] = res[1]
        out[r] = res
per_call = 2 * world
if mode == "ok":
    bad = [r for r in range(world) if out[r] != want]
    print("ranks equal to the one-context call:", not bad, "| stand-in all-gathers:", n_allgathers, "| want[3] =", want[3])
    ok = not bad and want[3] is True and n_allgathers == per_call * loops
else:
    print("codes:", codes, "| stand-in all-gathers:", n_allgathers)
    # a failure before exchange 1 ends the call there on every rank; one between the exchanges is carried by exchange 2; one
    # inside an exchange, before its gather (the injection sits in front of ncclAllGather), is carried by that same exchange,
    # which rank 0 enters a second time with its status
    calls = {"fail1": world, "fail2": 2 * world, "fail3": world, "fail4": 2 * world}[mode] * loops
    mine = pkg.ERR_NOMEM if mode in ("fail3", "fail4") else pkg.ERR_INVALID
    ok = codes[0] == mine and all(c == pkg.ERR_PEER for c in codes[1:]) and n_allgathers == calls
print("RCCL-STUB-RANKS-OK" if ok else "RCCL-STUB-RANKS-FAILED: %r" % (out,))
sys.stdout.flush()
os._exit(0 if ok else 1)
