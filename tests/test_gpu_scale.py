"""GPU parity at BASELINE.json's full sizes through size-independent properties: with bases k_i*G the
MSM must equal (sum k_i s_i)*G (one scalar mul validates 2^20 points), it must be linear in the scalars,
and a prefix MSM must match the CPU restatement of the reference algorithm on a bounded sample."""
import numpy as np
import pytest
import torch

from oracle import bn254 as O, cref

pytestmark = pytest.mark.gpu


def _workload(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.bytes(64 * n)
    vals = [int.from_bytes(raw[64 * i:64 * i + 64], "little") % O.R for i in range(n)]
    arr = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(n, 32)
    return vals, arr


@pytest.mark.parametrize("log2n", [16, 20])
def test_msm_identity_and_linearity(eng, log2n):
    n = 1 << log2n
    ks, k_np = _workload(n, 1)
    s1, s1_np = _workload(n, 2)
    s2, s2_np = _workload(n, 3)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        def msm(vals, arr):
            d = torch.from_numpy(arr.copy()).to(dev)
            torch.cuda.synchronize()
            return eng.g1_msm_device(table, d.data_ptr(), n)

        def kg(k):
            return eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1), O.fe_to_bytes(k % O.R)))

        r1, r2 = msm(s1, s1_np), msm(s2, s2_np)
        t1 = sum(k * s for k, s in zip(ks, s1)) % O.R
        t2 = sum(k * s for k, s in zip(ks, s2)) % O.R
        assert eng.g1_batch_to_affine(r1) == kg(t1) == O.aff_to_bytes(O.scalar_mul(t1, O.G1))
        assert eng.g1_batch_to_affine(r2) == kg(t2)
        # linearity: MSM(s1 + s2) == MSM(s1) + MSM(s2)
        s12 = [(a + b) % O.R for a, b in zip(s1, s2)]
        arr12 = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in s12), dtype=np.uint8).reshape(n, 32)
        r12 = msm(s12, arr12)
        assert eng.g1_batch_to_affine(r12) == eng.g1_batch_to_affine(eng.g1_batch_add(r1, r2))
        # bounded sample against the restated reference algorithm (mock/arith/ecc.rs:106-129)
        m = 4096
        bases = eng.bases_download(table, 0, m)
        assert bases[:64 * 8] == cref.g1_batch_to_affine(
            cref.g1_batch_scalar_mul(O.aff_to_bytes(O.G1) * 8, bytes(k_np[:8].tobytes()), 8), 8)
        got = eng.g1_batch_to_affine(eng.g1_msm_preloaded(table, bytes(s1_np[:m].tobytes())))
        assert got == cref.multi_exp_naive(bases, bytes(s1_np[:m].tobytes()), m)
    finally:
        eng.bases_free(table)


def test_msm_packed_sort_item_all_ones(eng):
    """Regression (found at 2^22 points with bench.py's seed): the packed sort item of the LAST point is
    0xFFFFFFFF when one of its digits is negative with sub-bucket SB-1 — it must not be taken for padding.
    Reproduced with a 19-bit index field: n = 2^19, c = 14, sub_bits = 12, last scalar crafted so that
    window 1 has raw digit 12288 (negative, magnitude 4096 -> bucket 4095 -> sub 4095)."""
    n, c = 1 << 19, 14
    ks, k_np = _workload(n, 21)
    ss, _ = _workload(n, 22)
    last = ss[-1]
    last &= ~((1 << (2 * c)) - 1)            # clear windows 0 and 1
    last |= 1 | (12288 << c)                  # window 0 = +1 (no carry), window 1 raw = 12288
    ss[-1] = last % O.R
    assert (ss[-1] >> c) & ((1 << c) - 1) == 12288
    s_np = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in ss), dtype=np.uint8).reshape(n, 32)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    d_s = torch.from_numpy(s_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    want_k = sum(k * s for k, s in zip(ks, ss)) % O.R
    want = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1), O.fe_to_bytes(want_k)))
    eng.msm_configure(window_bits=c)
    try:
        for sub_bits, tile in ((12, 0), (12, -2), (12, -1)):
            eng.msm_configure_sort(sub_bits, tile)
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want, (sub_bits, tile)
    finally:
        eng.msm_configure()
        eng.msm_configure_sort()
        eng.bases_free(table)
