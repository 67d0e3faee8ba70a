"""GPU: the multi-GPU exchange inside the C ABI (csrc/comm.inc: RCCL all-gather + local EC adds) and the RCCL path of
torch.distributed (backend "nccl" IS RCCL on ROCm).  The GPU box has ONE device, so every communicator here has one rank;
the multi-rank control flow is covered on CPU under gloo (tests/test_dist_gloo.py) and the >= 2-device cases are gated."""
import importlib
import os

import pytest
import torch

import __graft_entry__ as entry
from oracle import bn254 as O
from tests.util import to_jac_bytes

pytestmark = pytest.mark.gpu


def _points(seed, n):
    rng = O.SplitMix64(seed)
    return [O.scalar_mul(rng.fr(), O.G1) for _ in range(n)], rng


def test_single_process_group_one_device(pkg):
    """h2agg_comm_create([0]) -> ncclCommInitAll with one device; all-gather + fold of its own partials"""
    grp = pkg.H2AggGroup([0])
    try:
        assert grp.engines[0].comm_size() == 1
        pts, rng = _points(0xC0, 3)
        pts[1] = O.INF
        jac = to_jac_bytes(b"".join(O.aff_to_bytes(p) for p in pts), [rng.fr() or 1 for _ in pts])
        out = grp.allgather_add_points([jac])
        assert out == b"".join(O.aff_to_bytes(p) for p in pts)
        # the contexts are full engines
        assert grp.engines[0].g1_batch_to_affine(jac) == out
    finally:
        grp.close()


def test_rank_mode_unique_id_world_1(pkg):
    """one process per GPU: unique id -> h2agg_comm_init_rank(rank 0 of 1) -> h2agg_allgather_add_points(&ctx, 1, ...)"""
    eng = pkg.H2Agg(0)
    try:
        uid = pkg.H2Agg.comm_unique_id()
        assert len(uid) == 128
        eng.comm_init_rank(uid, 0, 1)
        assert eng.comm_size() == 1
        with pytest.raises(pkg.H2AggError):
            eng.comm_init_rank(uid, 0, 1)                        # a context holds one communicator
        pts, rng = _points(0xC1, 2)
        jac = to_jac_bytes(b"".join(O.aff_to_bytes(p) for p in pts), [rng.fr() or 1 for _ in pts])
        assert eng.allgather_add_points(jac) == b"".join(O.aff_to_bytes(p) for p in pts)
    finally:
        eng.close()


def test_allgather_without_communicator_is_an_error(eng, pkg):
    with pytest.raises(pkg.H2AggError):
        eng.allgather_add_points(pkg.IDENTITY_JAC)


def test_aggregate_sharded_over_c_abi_comm(pkg):
    """aggregate_sharded with the exchange done by the library (comm= the engine) instead of torch.distributed"""
    from tests.test_dist_gloo import make_proofs, reference_final_pair
    agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
    mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
    eng = pkg.H2Agg(0)
    try:
        eng.comm_init_rank(pkg.H2Agg.comm_unique_id(), 0, 1)
        backend = agg.GpuBackend(pkg, eng)
        lam = 0x1234567890ABCDEF1234567890ABCDEF % O.R
        n_total = 5
        pair = agg.aggregate_sharded(backend, lambda b, idx: make_proofs(mo, backend, b, idx, n_total), n_total,
                                     O.fe_to_bytes(lam), comm=eng)
        assert pair[0] + pair[1] == reference_final_pair(n_total, lam)
    finally:
        eng.close()


def test_torch_nccl_world_1_all_gather(eng, pkg):
    """bench.py's N > 1 exchange (`init_process_group("nccl")` + all_gather of the 96-byte accumulators, then the local
    fold) executed with world_size 1, so that the RCCL code path runs at least once under the driver"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29621"
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        pts, _ = _points(0xC2, 1)
        jac = to_jac_bytes(O.aff_to_bytes(pts[0]), [7])
        mine = torch.frombuffer(bytearray(jac), dtype=torch.uint8).to(dev)
        gathered = [torch.empty_like(mine)]
        dist.all_gather(gathered, mine)
        dist.barrier()
        torch.cuda.synchronize(dev)
        folded = eng.g1_sum(bytes(torch.stack(gathered).cpu().numpy().tobytes()))
        assert eng.g1_batch_to_affine(folded) == O.aff_to_bytes(pts[0])
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_single_process_group_two_devices(pkg):
    grp = pkg.H2AggGroup([0, 1])
    try:
        pts, rng = _points(0xC3, 4)
        a = to_jac_bytes(b"".join(O.aff_to_bytes(p) for p in pts[:2]), [3, 5])
        b = to_jac_bytes(b"".join(O.aff_to_bytes(p) for p in pts[2:]), [7, 11])
        out = grp.allgather_add_points([a, b])
        assert out == O.aff_to_bytes(O.add(pts[0], pts[2])) + O.aff_to_bytes(O.add(pts[1], pts[3]))
    finally:
        grp.close()
