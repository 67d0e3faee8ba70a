"""Sharded aggregation FROM PROOF BYTES with the lambda exchange (exchange 1 + exchange 2;
halo2-snark-aggregator-api/src/systems/halo2/verify.rs:909-913, :924-938).

CPU part (this file's non-gpu tests): the oracle's restatement of ONE RANK of the sharded algorithm
(oracle/verifier.py::verify_aggregation_sharded_rank) run as world-size-2 / -3 `gloo` process groups must give, on every
rank, the pair and lambda of the oracle's one-process verify_aggregation_proofs_in_chip on the same trapdoor proofs.
The GPU part (tests/test_gpu_sharded.py) holds the product (h2agg_verify_aggregation_sharded) to the same."""
import multiprocessing as mp
import os

import pytest

from oracle import bn254 as O
from oracle import schema as S
from oracle import verifier as V
from tests.test_verifier_pipeline import SHAPES, make_batch


def shard_batch(circuits, world, rank):
    """round-robin over the aggregation order (circuits in order, proofs in order) -> (local circuits, global indices)"""
    local, gidx, g = [], [], 0
    for c in circuits:
        lc = V.CircuitProofs(c.name, c.cs, c.g_lagrange)
        for pr in c.proofs:
            if g % world == rank:
                lc.proofs.append(pr)
                gidx.append(g)
            g += 1
        local.append(lc)
    return local, gidx, g


def _worker(rank, world, port, seed, shape_ids, nproofs, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import __graft_entry__ as entry
        import importlib
        entry.load_package()
        ver = importlib.import_module(entry.PKG_NAME + ".verifier")   # (dist_allgather is transport glue, no compute)
        _setup, circuits = make_batch(seed, [SHAPES[i] for i in shape_ids], nproofs)
        local, gidx, n_total = shard_batch(circuits, world, rank)
        left, right, lam = V.verify_aggregation_sharded_rank(S.OracleEccChip(), local, gidx, n_total, ver.dist_allgather(dist))
        q.put((rank, S.final_pair_bytes(left, right), lam))
        dist.destroy_process_group()
    except BaseException as e:   # noqa
        import traceback
        q.put((rank, "ERR " + traceback.format_exc(), None))


@pytest.mark.parametrize("world,shape_ids,nproofs", [(2, (0,), 3), (2, (0, 1), 2), (3, (0,), 2)])
def test_sharded_from_bytes_world_gloo_matches_one_process(world, shape_ids, nproofs):
    seed = 0x5A0 + world * 16 + len(shape_ids) * 4 + nproofs
    _setup, circuits = make_batch(seed, [SHAPES[i] for i in shape_ids], nproofs)
    want_l, want_r, _plain, _commits, want_lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 13 + nproofs) % 2000
    ps = [ctx.Process(target=_worker, args=(r, world, port, seed, shape_ids, nproofs, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for rank, pair, lam in got:
        assert not (isinstance(pair, str) and pair.startswith("ERR")), pair
        assert lam == want_lam, "rank %d derives another lambda" % rank
        assert pair == S.final_pair_bytes(want_l, want_r), "rank %d: final pair" % rank


def test_shard_batch_covers_every_position_once():
    _setup, circuits = make_batch(0x5B1, [SHAPES[0], SHAPES[2]], 3)
    seen = []
    for r in range(4):
        _local, gidx, n = shard_batch(circuits, 4, r)
        seen += gidx
    assert sorted(seen) == list(range(n)) and n == 6


def test_rank_function_single_rank_is_the_reference_fold():
    """world 1: the expanded fold sum lambda^(N-1-i) proof_i equals the nested acc = acc * lambda + proof (verify.rs:926-938)"""
    _setup, circuits = make_batch(0x5C2, [SHAPES[1]], 3)
    want_l, want_r, _p, _c, want_lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    local, gidx, n = shard_batch(circuits, 1, 0)
    left, right, lam = V.verify_aggregation_sharded_rank(S.OracleEccChip(), local, gidx, n, lambda b: [b])
    assert (left, right, lam) == (want_l, want_r, want_lam)
