"""GPU: h2agg_verify_aggregation_sharded — the aggregation sharded over ranks FROM PROOF BYTES, with both exchanges inside the
C ABI (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:909-913, :924-938).

Every rank must return the pair, lambda and pairing verdict of the one-context h2agg_verify_aggregation on all proofs, bit for
bit.  Ranks here are (a) threads of this process, one context each on cuda:0, exchanging through an in-process barrier — the
`allgather` callback of the C ABI, i.e. the host's own transport — and (b) two processes on cuda:0 exchanging over a `gloo`
group.  (RCCL refuses two ranks on one device; the RCCL transport of the same entry point runs at world 1 here, at world 2 / 3 over
the stream-ordered stand-in in tests/test_gpu_zz_standins.py — which collects LAST, so that a failure of test infrastructure
cannot hide parity tests under `pytest -x` — and at world N in bench.py --gpus N.)"""
import importlib
import multiprocessing as mp
import os
import threading

import pytest

import __graft_entry__ as entry
from oracle import bn254 as O
from oracle import schema as S
from oracle import verifier as V
from tests.test_gpu_verifier import run_product
from tests.test_pairing_capi import g2b
from tests.test_sharded_from_bytes import shard_batch
from tests.test_verifier_pipeline import SHAPES, make_batch

pytestmark = pytest.mark.gpu


def product_args(ver, eng, setup, circuits_local):
    table = eng.bases_upload(b"".join(O.aff_to_bytes(p) for p in setup.g_lagrange))
    vks, arg = [], []
    for c in circuits_local:
        vk = ver.VerifyingKey(eng, ver.encode_vk(c.cs, O.aff_to_bytes))
        vks.append(vk)
        arg.append((vk, c.name, table, [([b"".join(O.fe_to_bytes(v) for v in col) for col in inst[0]], data) for inst, data in c.proofs]))
    return table, vks, arg


class BarrierExchange:
    """all-gather between the threads of one process"""

    def __init__(self, world):
        self.world, self.slots, self.bar = world, [None] * world, threading.Barrier(world)

    def for_rank(self, rank):
        def allgather(payload):
            self.slots[rank] = payload
            self.bar.wait(timeout=120)
            out = list(self.slots)
            self.bar.wait(timeout=120)
            return out
        return allgather


def run_threads(pkg, setup, circuits, world, backend):
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    ex = BarrierExchange(world)
    results, errors = [None] * world, []

    def rank_main(rank):
        eng = pkg.H2Agg(0)
        try:
            eng.transcript_configure(backend)
            local, gidx, n_total = shard_batch(circuits, world, rank)
            table, vks, arg = product_args(ver, eng, setup, local)
            try:
                results[rank] = ver.verify_aggregation_sharded(eng, arg, gidx, n_total, rank, world, ex.for_rank(rank),
                                                               g2b(setup.s_g2), g2b(setup.g2))
            finally:
                for vk in vks:
                    vk.close()
                eng.bases_free(table)
        except BaseException as e:   # noqa
            errors.append((rank, e))
            ex.bar.abort()
        finally:
            eng.close()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errors, errors
    return results


@pytest.mark.parametrize("backend", ["device", "host"])
@pytest.mark.parametrize("world,shape_ids,nproofs", [(2, (0,), 3), (2, (0, 1, 2), 2), (3, (1,), 2)])
def test_sharded_threads_equal_the_one_context_call(eng, pkg, backend, world, shape_ids, nproofs):
    setup, circuits = make_batch(0x6A0 + world * 16 + len(shape_ids) * 4 + nproofs, [SHAPES[i] for i in shape_ids], nproofs)
    eng.transcript_configure(backend)
    try:
        want = run_product(pkg, eng, setup, circuits)            # (left, right, lambda, pairing_ok) of ONE context
    finally:
        eng.transcript_configure("auto")
    assert want[3] is True
    # ... which the oracle confirms
    ol, orr, _p, _c, olam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    assert want[0] + want[1] == S.final_pair_bytes(ol, orr) and want[2] == O.fe_to_bytes(olam)
    got = run_threads(pkg, setup, circuits, world, backend)      # world 3 with 2 proofs: rank 2 holds none
    for rank, res in enumerate(got):
        assert res == want, "rank %d differs from the one-context aggregation" % rank


def test_sharded_world_1_is_the_plain_call(eng, pkg):
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    setup, circuits = make_batch(0x6B7, [SHAPES[0]], 2)
    want = run_product(pkg, eng, setup, circuits)
    table, vks, arg = product_args(ver, eng, setup, circuits)
    try:
        got = ver.verify_aggregation_sharded(eng, arg, [0, 1], 2, 0, 1, lambda b: [b], g2b(setup.s_g2), g2b(setup.g2))
    finally:
        for vk in vks:
            vk.close()
        eng.bases_free(table)
    assert got == want


def test_two_ranks_claiming_one_position_is_refused(eng, pkg):
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    setup, circuits = make_batch(0x6C3, [SHAPES[0]], 2)
    table, vks, arg = product_args(ver, eng, setup, circuits)
    try:
        with pytest.raises(Exception):   # both "ranks" (the echo below) claim positions 0 and 1
            ver.verify_aggregation_sharded(eng, arg, [0, 1], 2, 0, 2, lambda b: [b, b])
        with pytest.raises(Exception):   # position 1 is nobody's
            one = [(arg[0][0], arg[0][1], arg[0][2], arg[0][3][:1])]
            ver.verify_aggregation_sharded(eng, one, [0], 2, 0, 1, lambda b: [b])
        with pytest.raises(Exception):   # no transport and no communicator
            ver.verify_aggregation_sharded(eng, arg, [0, 1], 2, 0, 2, None)
    finally:
        for vk in vks:
            vk.close()
        eng.bases_free(table)


def _proc(rank, world, port, seed, q):
    try:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = entry.load_package()
        ver = importlib.import_module(entry.PKG_NAME + ".verifier")
        eng = pkg.H2Agg(0)
        setup, circuits = make_batch(seed, [SHAPES[0], SHAPES[2]], 2)
        local, gidx, n_total = shard_batch(circuits, world, rank)
        table, vks, arg = product_args(ver, eng, setup, local)
        res = ver.verify_aggregation_sharded(eng, arg, gidx, n_total, rank, world, ver.dist_allgather(dist), g2b(setup.s_g2), g2b(setup.g2))
        q.put((rank, res))
        dist.destroy_process_group()
    except BaseException:   # noqa
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


def test_two_processes_on_one_gpu_over_gloo(eng, pkg):
    seed = 0x6D9
    setup, circuits = make_batch(seed, [SHAPES[0], SHAPES[2]], 2)
    want = run_product(pkg, eng, setup, circuits)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_proc, args=(r, 2, port, seed, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=900) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for rank, res in got:
        assert not isinstance(res, str), res
        assert res == want, "rank %d differs from the one-context aggregation" % rank


def test_sharded_over_the_contexts_rccl_communicator_world_1(pkg, eng):
    """transport = the context's own RCCL communicator (shard->allgather NULL): ncclAllGather of both exchanges inside the C ABI"""
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    setup, circuits = make_batch(0x6E4, [SHAPES[1]], 2)
    want = run_product(pkg, eng, setup, circuits)
    e2 = pkg.H2Agg(0)
    try:
        e2.comm_init_rank(pkg.H2Agg.comm_unique_id(), 0, 1)
        table, vks, arg = product_args(ver, e2, setup, circuits)
        try:
            got = ver.verify_aggregation_sharded(e2, arg, [0, 1], 2, 0, 1, None, g2b(setup.s_g2), g2b(setup.g2))
        finally:
            for vk in vks:
                vk.close()
            e2.bases_free(table)
    finally:
        e2.close()
    assert got == want


def run_threads_collecting(pkg, setup, circuits, world, prepare=None, wrap_exchange=None):
    """as run_threads, but every rank's outcome is kept: a result tuple, or the H2AggError it raised.  prepare(rank, eng):
    per-rank set-up (debug keys); wrap_exchange(rank, allgather) -> allgather: a transport that misbehaves."""
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    ex = BarrierExchange(world)
    out = [None] * world

    def rank_main(rank):
        eng = pkg.H2Agg(0)
        try:
            if prepare:
                prepare(rank, eng)
            local, gidx, n_total = shard_batch(circuits, world, rank)
            table, vks, arg = product_args(ver, eng, setup, local)
            try:
                ag = ex.for_rank(rank)
                if wrap_exchange:
                    ag = wrap_exchange(rank, ag)
                out[rank] = ver.verify_aggregation_sharded(eng, arg, gidx, n_total, rank, world, ag, g2b(setup.s_g2), g2b(setup.g2))
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
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank is still waiting in an exchange another rank never entered"
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_failing_before_the_lambda_exchange_hangs_nobody(eng, pkg, world):
    """One malformed proof on one rank (a point that does not decode) used to leave the other ranks in an all-gather forever
    (ADVICE r4): now the failing rank still enters exchange 1, with its error in the status word; it reports its own error and
    every other rank H2AGG_ERR_PEER."""
    setup, circuits = make_batch(0x6F1 + world, [SHAPES[0]], 3)
    inst, data = circuits[0].proofs[1]          # round-robin: proof 1 sits on rank 1
    bad = bytearray(data)
    x = next(v for v in range(2, 60) if pow((v ** 3 + 3) % O.P, (O.P - 1) // 2, O.P) != 1)
    bad[0:32] = x.to_bytes(32, "little")
    circuits[0].proofs[1] = (inst, bytes(bad))
    out = run_threads_collecting(pkg, setup, circuits, world)
    for rank, res in enumerate(out):
        assert isinstance(res, pkg.H2AggError), (rank, res)
        assert res.code == (pkg.ERR_BAD_POINT if rank == 1 else pkg.ERR_PEER), (rank, res)


@pytest.mark.parametrize("phase", [1, 2])
def test_an_injected_failure_on_either_side_of_the_lambda_exchange(eng, pkg, phase):
    """debug key shard_fail: rank 0 fails before (1) / between (2) the exchanges — exchange 2 must then carry the status"""
    setup, circuits = make_batch(0x6F8, [SHAPES[1]], 2)
    out = run_threads_collecting(pkg, setup, circuits, 2, prepare=lambda rank, e: e.debug_configure("shard_fail", phase if rank == 0 else 0))
    assert isinstance(out[0], pkg.H2AggError) and out[0].code == pkg.ERR_INVALID, out[0]
    assert isinstance(out[1], pkg.H2AggError) and out[1].code == pkg.ERR_PEER, out[1]
    assert "rank 0" in str(out[1])
    # and the contexts are fine afterwards: the same aggregation without the injection
    want = run_product(pkg, eng, setup, circuits)
    assert run_threads(pkg, setup, circuits, 2, "auto") == [want, want]


def test_a_partial_pair_off_the_curve_is_refused(eng, pkg):
    """what arrives in exchange 2 is checked like any input: canonical coordinates of a point ON the curve"""
    setup, circuits = make_batch(0x6FB, [SHAPES[0]], 2)

    def wrap(rank, ag):
        def allgather(payload):
            parts = ag(payload)
            if len(payload) == 132:                       # exchange 2: rank 1's W_x partial with y + 1
                p = bytearray(parts[1])
                y = (int.from_bytes(p[32:64], "little") + 1) % O.P
                p[32:64] = y.to_bytes(32, "little")
                parts = [parts[0], bytes(p)]
            return parts
        return allgather
    out = run_threads_collecting(pkg, setup, circuits, 2, wrap_exchange=wrap)
    for rank, res in enumerate(out):
        assert isinstance(res, pkg.H2AggError) and res.code == pkg.ERR_BAD_POINT, (rank, res)
