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
