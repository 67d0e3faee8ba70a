"""GPU: the sharded-aggregation host path on the product backend (aggregate.GpuBackend -> libh2agg.so)."""
import importlib

import pytest

import __graft_entry__ as entry
from oracle import bn254 as O
from tests.test_dist_gloo import make_proofs, reference_final_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_total,world", [(5, 1), (5, 2), (6, 4)])
def test_sharded_aggregation_gpu_backend(eng, pkg, n_total, world):
    agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
    mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
    backend = agg.GpuBackend(pkg, eng)
    lam = 0x1234567890ABCDEF1234567890ABCDEF % O.R
    lefts, rights = [], []
    for rank in range(world):                       # every shard on this one GPU, then the fold the ranks would do
        idx = agg.shard_indices(n_total, world, rank)
        b = backend.new_builder()
        proofs = make_proofs(mo, backend, b, idx, n_total)
        local = agg.local_weighted_proof(b, proofs, idx, n_total, O.fe_to_bytes(lam))
        if local is None:
            l, r = agg.IDENTITY_AFF, agg.IDENTITY_AFF
        else:
            l, r = backend.evaluate(b, local)
        lefts.append(l)
        rights.append(r)
        b.close()
    got = backend.sum_affine(lefts) + backend.sum_affine(rights)
    assert got == reference_final_pair(n_total, lam)


@pytest.mark.parametrize("n,world", [(1000, 1), (1000, 3), (5, 8)])
def test_msm_split_by_points_gpu_backend(eng, pkg, n, world):
    """aggregate.msm_sliced's per-rank work on the product backend: every slice's MSM on this one GPU, then the fold the
    ranks would do after the all-gather; equals the unsplit MSM (and the oracle)."""
    agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
    backend = agg.GpuBackend(pkg, eng)
    rng = O.SplitMix64(0x51CE + n)
    ks = [rng.fr() for _ in range(n)]
    bases = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1) * n, b"".join(O.fe_to_bytes(k) for k in ks)))
    scs = [rng.fr() for _ in range(n)]
    scalars = b"".join(O.fe_to_bytes(s) for s in scs)
    parts = []
    for rank in range(world):
        lo, hi = agg.slice_bounds(n, world, rank)
        parts.append(backend.msm(bases[64 * lo:64 * hi], scalars[32 * lo:32 * hi]) if hi > lo else agg.IDENTITY_AFF)
    got = backend.sum_affine(parts)
    assert got == backend.msm(bases, scalars) == agg.msm_sliced(backend, bases, scalars)
    assert got == O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, scs)) % O.R, O.G1))
