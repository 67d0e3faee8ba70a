"""Child process of tests/test_gpu_sharded.py::test_sharded_over_rccl_transport_with_threads_as_ranks (not collected by pytest).

    python tests/rccl_stub_ranks.py <librccl.so.1 stand-in> <world> <ok | fail1 | fail2>

h2agg_verify_aggregation_sharded with shard->allgather = NULL — the library's OWN transport: h2agg_comm_init_rank, then both
exchanges as ncclAllGather on the context's stream between a host-to-device and a device-to-host copy (csrc/verifier.inc
shard_allgather) — at world > 1 on a one-GPU box: the ranks are threads of this process, one context each, and RCCL is
tests/cpp/rccl_stub.cpp in its threads-as-ranks mode, loaded here BEFORE libh2agg.so resolves RCCL (this process never imports
torch, whose wheel carries the real one).  Every rank must return the one-context call's pair, lambda and verdict; with an
injected failure on rank 0 (debug key shard_fail) every rank must return — rank 0 its own error, the others H2AGG_ERR_PEER."""
import ctypes
import importlib
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stub, world, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
_keep = ctypes.CDLL(stub, mode=ctypes.RTLD_GLOBAL)

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
eng0 = pkg.H2Agg(0)
want = run_product(pkg, eng0, setup, circuits)
uid = pkg.H2Agg.comm_unique_id()
out = [None] * world


def rank_main(rank):
    eng = pkg.H2Agg(0)
    try:
        eng.comm_init_rank(uid, rank, world)
        assert eng.comm_size() == world and eng.comm_rank() == rank
        if mode != "ok" and rank == 0:
            eng.debug_configure("shard_fail", 1 if mode == "fail1" else 2)
        local, gidx, n_total = shard_batch(circuits, world, rank)
        table, vks, arg = product_args(ver, eng, setup, local)
        try:
            out[rank] = ver.verify_aggregation_sharded(eng, arg, gidx, n_total, rank, world, None, g2b(setup.s_g2), g2b(setup.g2))
        finally:
            for vk in vks:
                vk.close()
            eng.bases_free(table)
    except BaseException as e:   # noqa
        out[rank] = e
    finally:
        eng.close()


ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
for t in ts:
    t.start()
for t in ts:
    t.join(timeout=240)
if any(t.is_alive() for t in ts):
    print("HANG: a rank is still inside an exchange", flush=True)
    os._exit(3)
n_allgathers = _keep.rccl_stub_allgathers()
if mode == "ok":
    bad = [r for r in range(world) if out[r] != want]
    print("ranks equal to the one-context call:", not bad, "| stub all-gathers:", n_allgathers, "| want[3] =", want[3])
    ok = not bad and want[3] is True and n_allgathers == 2 * world
else:
    codes = [getattr(o, "code", None) for o in out]
    print("codes:", codes, "| stub all-gathers:", n_allgathers)
    # (a failure before exchange 1 ends the call there on every rank; one between the exchanges is carried by exchange 2)
    ok = codes[0] == pkg.ERR_INVALID and all(c == pkg.ERR_PEER for c in codes[1:]) and n_allgathers == (world if mode == "fail1" else 2 * world)
print("RCCL-STUB-RANKS-OK" if ok else "RCCL-STUB-RANKS-FAILED: %r" % (out,))
sys.stdout.flush()
os._exit(0 if ok else 1)
