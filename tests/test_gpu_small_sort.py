"""GPU: the one-launch sort of small MSMs (`k_small_sort`, csrc/sort_kernels.hpp) and the split MSM built on it — the two
multi_exps of EvaluationQuerySchema-based `evaluate_multiopen_proof` (halo2-snark-aggregator-api/src/systems/halo2/multiopen.rs,
arith/ecc.rs:38-58 `multi_exp`) as ONE set of launches over the concatenated (point, scalar) pairs.

Expected values: (sum k_i s_i) G from the Python oracle; the packed two-level sort (`small_sort` = 0) and two separate MSMs
(`eval_split` = 0) must agree bit for bit."""
import numpy as np
import pytest
import torch

from oracle import bn254 as O
from oracle import schema as S
from tests.util import fr_bytes

pytestmark = pytest.mark.gpu


def _vals(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.bytes(64 * n)
    return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % O.R for i in range(n)]


def _bases(eng, ks):
    n = len(ks)
    arr = np.frombuffer(fr_bytes(ks), dtype=np.uint8).reshape(n, 32)
    d_k = torch.from_numpy(arr.copy()).to(torch.device("cuda", 0))
    t = eng.bases_generate(d_k.data_ptr(), n)
    try:
        return eng.bases_download(t, 0, n)
    finally:
        eng.bases_free(t)


def _want(ks, ss):
    return O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % O.R, O.G1))


@pytest.fixture(autouse=True)
def _restore(eng):
    yield
    eng.debug_configure("small_sort", 1)
    eng.debug_configure("eval_split", 1)
    eng.msm_configure()
    eng.msm_configure_glv(0)


@pytest.mark.parametrize("glv", [1, -1])
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 1000, 1025, 4097, 16384, 16385])
def test_small_sort_sizes(eng, n, glv):
    """every size class around the workgroup (1024 threads), the window table's steps and the path's upper limit (16384; one
    more point takes the packed sort)"""
    ks, ss = _vals(n, 9000 + n), _vals(n, 9500 + n)
    if n > 4:
        ss[1] = 0                                # a scalar without digits
        ss[2] = O.R - 1                          # every digit at the recoding's edge
        ks[3], ss[3] = ks[0], ss[0]              # the same pair twice: P + P inside a bucket
    bases, sb, want = _bases(eng, ks), fr_bytes(ss), _want(ks, ss)
    eng.msm_configure_glv(glv)
    got = eng.g1_batch_to_affine(eng.g1_msm(bases, sb))
    assert got == want
    eng.debug_configure("small_sort", 0)
    assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want


@pytest.mark.parametrize("glv", [1, -1])
@pytest.mark.parametrize("c", [2, 5, 8, 11, 13, 14, 15])
def test_small_sort_window_widths(eng, c, glv):
    """forced widths: 2 buckets per window up to the 8192 the kernel keeps in LDS (c = 14); c = 15 falls back to the packed sort"""
    n = 3000
    ks, ss = _vals(n, 9100 + c), _vals(n, 9150 + c)
    bases, sb, want = _bases(eng, ks), fr_bytes(ss), _want(ks, ss)
    eng.msm_configure(window_bits=c)
    eng.msm_configure_glv(glv)
    assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want


@pytest.mark.parametrize("kind", ["equal", "small", "sparse", "zero"])
def test_small_sort_skewed_scalars(eng, kind):
    """one over-long bucket per window (the chunked path behind the sort), empty upper windows, mostly empty buckets, nothing"""
    n = 5000
    ks, ss = _vals(n, 9200), _vals(n, 9201)
    if kind == "equal":
        ss = [ss[0]] * n
    elif kind == "small":
        ss = [s % 1000 for s in ss]
    elif kind == "sparse":
        ss = [s if i % 7 == 0 else 0 for i, s in enumerate(ss)]
    else:
        ss = [0] * n
    bases, sb = _bases(eng, ks), fr_bytes(ss)
    got = eng.g1_msm(bases, sb)
    if kind == "zero":
        assert got[64:96] == bytes(32)           # Jacobian z = 0: the identity
        return
    assert eng.g1_batch_to_affine(got) == _want(ks, ss)


@pytest.mark.parametrize("n,batch,glv", [(77, 20, 0), (3000, 5, 1), (3000, 5, -1), (16384, 1, 0), (8192, 2, -1)])
def test_small_sort_batches_over_one_table(eng, n, batch, glv):
    """`batch` MSMs over one table: scalar j belongs to MSM j / n (batches of more than 16384 scalars in all take the packed
    sort — both must give the single MSMs' results)"""
    dev = torch.device("cuda", 0)
    ks = _vals(n, 9300)
    k_np = np.frombuffer(fr_bytes(ks), dtype=np.uint8).reshape(n, 32)
    table = eng.bases_generate(torch.from_numpy(k_np.copy()).to(dev).data_ptr(), n)
    eng.msm_configure_glv(glv)
    try:
        rows = [_vals(n, 9310 + q) for q in range(batch)]
        if batch > 1:
            rows[1] = [0] * n
        d_s = torch.from_numpy(np.stack([np.frombuffer(fr_bytes(r), dtype=np.uint8).reshape(n, 32) for r in rows]).copy()).to(dev)
        d_out = torch.zeros((batch, 96), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.g1_msm_device_batch_async(table, d_s.data_ptr(), n, batch, d_out.data_ptr())
        got = eng.g1_batch_to_affine_device(d_out.data_ptr(), batch)
        for q in range(batch):
            t = sum(k * s for k, s in zip(ks, rows[q])) % O.R
            assert got[64 * q:64 * q + 64] == O.aff_to_bytes(O.scalar_mul(t, O.G1)), q
    finally:
        eng.bases_free(table)


def _fold(pkg, eng, rng, nproofs, ncommit):
    """`nproofs` synthetic multi-open proofs folded with lambda (verify.rs:926-938): the oracle's pair and the builder's sides"""
    want_proofs, got = [], []
    b = pkg.SchemaBuilder(eng)
    for i in range(nproofs):
        key = "s_p%d" % i
        x = rng.fr()
        rp = {0: x, 1: x * 7 % O.R, -6: x * 11 % O.R}
        rots = [0] * ncommit + [1, 0, -6, 1, 0, -6, 0]
        spec = [(rot, "%s_q%d" % (key, k), rp[rot], O.scalar_mul(rng.fr(), O.G1), rng.fr()) for k, rot in enumerate(rots)]
        w = [O.scalar_mul(rng.fr(), O.G1) for _ in range(3)]
        v, u = rng.fr(), rng.fr()
        want_proofs.append(S.batch_multi_open_proofs(key, [S.evaluation_query(*q) for q in spec], w, v, u))
        qn = b.evaluation_queries([k for _r, k, _z, _c, _e in spec], b"".join(O.aff_to_bytes(c) for *_x, c, _e in spec),
                                  b"".join(O.fe_to_bytes(e) for *_x, e in spec))
        got.append(b.batch_multi_open(key, [r for r, *_x in spec], b"".join(O.fe_to_bytes(z) for _r, _k, z, _c, _e in spec),
                                      qn, b"".join(O.aff_to_bytes(p) for p in w), O.fe_to_bytes(v), O.fe_to_bytes(u)))
    lam = rng.fr()
    agg = S.aggregate_fold(want_proofs, lam)
    lam_b = O.fe_to_bytes(lam)
    acc_x, acc_g = got[0]
    for w_x, w_g in got[1:]:
        acc_x, acc_g = acc_x * b.scalar(lam_b) + w_x, acc_g * b.scalar(lam_b) + w_g
    return b, agg, acc_x, acc_g


@pytest.mark.parametrize("nproofs,ncommit", [(1, 2), (3, 5), (4, 40)])
def test_evaluation_sides_as_one_split_msm(eng, pkg, nproofs, ncommit):
    """evaluate_multiopen_proof: both multi_exps in one set of launches (default) = two MSMs = the oracle's pair"""
    b, agg, acc_x, acc_g = _fold(pkg, eng, O.SplitMix64(0x5B17 + nproofs), nproofs, ncommit)
    want_l, want_r, want_names = S.evaluate_multiopen_proof(S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip(), agg)
    for split in (1, 0, 1):
        eng.debug_configure("eval_split", split)
        left, right, names = b.evaluate_multiopen_proof(acc_x, acc_g)
        assert left + right == S.final_pair_bytes(want_l, want_r) and names == want_names, split
    eng.debug_configure("small_sort", 0)         # without the one-launch sort there is no split MSM either
    left, right, names = b.evaluate_multiopen_proof(acc_x, acc_g)
    assert left + right == S.final_pair_bytes(want_l, want_r)
