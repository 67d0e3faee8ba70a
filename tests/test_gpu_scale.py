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


# Python model of csrc/sort_kernels.hpp glv_decompose (same constants), used to craft scalars
_LAM = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
_A1, _B1N = 147946756881789319000765030803803410728, 9931322734385697763
_A2, _B2 = 9931322734385697763, 147946756881789319010696353538189108491
_G1, _G2 = 0x24ccef014a773d2d25398fd0300ff6565, 0x2d91d232ec7e0b3d7


def glv_model(k):
    c1 = (k * _G1 + (1 << 255)) >> 256
    c2 = (k * _G2 + (1 << 255)) >> 256
    return k - c1 * _A1 - c2 * _A2, c1 * _B1N - c2 * _B2


def test_glv_model_and_extremes(eng):
    """The endomorphism split must be exact for every scalar: k1 + lambda*k2 = k (mod r), |k_i| < 2^127.
    Checked on the model for structured values, and on the GPU through MSM parity with GLV on vs off."""
    special = [0, 1, 2, O.R - 1, O.R - 2, _LAM, O.R - _LAM, (_LAM * _LAM) % O.R, (O.R - 1) // 2, (O.R + 1) // 2,
               (1 << 127) - 1, 1 << 127, (1 << 128) - 1, 1 << 253, _A1, _B2, O.R - _A1, (1 << 254) % O.R]
    rng = O.SplitMix64(31)
    special += [rng.fr() for _ in range(2000)]
    for k in special:
        k1, k2 = glv_model(k)
        assert (k1 + k2 * _LAM - k) % O.R == 0 and abs(k1) < (1 << 127) and abs(k2) < (1 << 127)
    n = len(special)
    ks = [rng.fr() for _ in range(n)]
    from tests.util import points_from_scalars, fr_bytes
    bases = points_from_scalars(ks)
    want = O.aff_to_bytes(O.scalar_mul(sum(a * b for a, b in zip(ks, special)) % O.R, O.G1))
    try:
        for mode in (1, -1):
            eng.msm_configure_glv(mode)
            for c in (0, 7, 16):
                eng.msm_configure(window_bits=c)
                assert eng.g1_batch_to_affine(eng.g1_msm(bases, fr_bytes(special))) == want, (mode, c)
    finally:
        eng.msm_configure()
        eng.msm_configure_glv(0)


def test_msm_packed_sort_item_all_ones(eng):
    """Regression (found at 2^22 points with bench.py's seed): the packed sort item of the LAST point can be
    0xFFFFFFFF — sub-bucket SB-1, negative, endo, index 2^idx_bits - 1 — and must not be taken for padding.
    Reproduced with an 18-bit index field: n = 2^18, c = 14, sub_bits = 12; the last scalar is built as
    k1 + lambda*k2 with k2's window-1 digit raw = 12288 (negative, magnitude 4096 -> bucket 4095 -> sub 4095)
    and the model confirms the device decomposition returns exactly (k1, k2)."""
    n, c = 1 << 18, 14
    ks, k_np = _workload(n, 21)
    ss, _ = _workload(n, 22)
    k1 = 5
    k2 = 1 | (12288 << c) | (3 << (2 * c))      # window 0 = +1 (no carry), window 1 raw = 12288, window 2 = 3 (+1 carry)
    last = (k1 + _LAM * k2) % O.R
    assert glv_model(last) == (k1, k2)
    ss[-1] = last
    s_np = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in ss), dtype=np.uint8).reshape(n, 32)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    d_s = torch.from_numpy(s_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    want_k = sum(k * s for k, s in zip(ks, ss)) % O.R
    want = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1), O.fe_to_bytes(want_k)))
    eng.msm_configure(window_bits=c)
    eng.msm_configure_glv(1)
    try:
        for sub_bits, tile in ((12, 0), (12, -2), (12, -1)):
            eng.msm_configure_sort(sub_bits, tile)
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want, (sub_bits, tile)
    finally:
        eng.msm_configure()
        eng.msm_configure_sort()
        eng.msm_configure_glv(0)
        eng.bases_free(table)


def test_msm_packed_sort_item_all_ones_plain(eng):
    """Same regression without GLV (19-bit index field, no endo bit): n = 2^19, c = 14, sub_bits = 12, the last scalar's
    window 1 has raw digit 12288 (negative, magnitude 4096 -> bucket 4095 -> sub 4095) -> item 0xFFFFFFFF."""
    n, c = 1 << 19, 14
    ks, k_np = _workload(n, 23)
    ss, _ = _workload(n, 24)
    last = ss[-1] & ~((1 << (2 * c)) - 1)
    last |= 1 | (12288 << c)
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
    eng.msm_configure_glv(-1)
    try:
        for sub_bits, tile in ((12, 0), (12, -2), (12, -1)):
            eng.msm_configure_sort(sub_bits, tile)
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want, (sub_bits, tile)
    finally:
        eng.msm_configure()
        eng.msm_configure_sort()
        eng.msm_configure_glv(0)
        eng.bases_free(table)


def test_msm_2p20_against_cpu_pippenger(eng):
    """Full-size cross-check against an independent implementation: the oracle's multi-threaded CPU Pippenger
    (4x64-bit Montgomery, Jacobian, unsigned windows) on all 2^20 points — not only the (sum k_i s_i) G identity."""
    n = 1 << 20
    _, k_np = _workload(n, 41)
    _, s_np = _workload(n, 42)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    d_s = torch.from_numpy(s_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        got = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
        bases = eng.bases_download(table, 0, n)
        assert got == cref.msm_pippenger(bases, bytes(s_np.tobytes()), n, 16, 16)
    finally:
        eng.bases_free(table)


@pytest.mark.parametrize("n,batch,glv", [(3000, 5, 0), (1 << 15, 3, 0), (1 << 15, 3, -1), (77, 20, 0), (1 << 17, 6, 0),
                                          (1 << 15, 19, 0), (1 << 14, 70, -1)])
def test_msm_batch_over_one_table(eng, n, batch, glv):
    """h2agg_g1_msm_device_batch_async: `batch` MSMs over the same bases in one set of launches must each equal
    the single MSM (and (sum k_i s_i) * G); covers chunking (19 x 2^15 and 70 x 2^14 are split into several sets of
    launches) and both scalar recodings."""
    dev = torch.device("cuda", 0)
    ks, k_np = _workload(n, 11)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    eng.msm_configure_glv(glv)
    try:
        rows, arrs = [], []
        for q in range(batch):
            v, a = _workload(n, 100 + q)
            if q == 1:                       # an all-zero scalar vector inside the batch: identity result
                v, a = [0] * n, np.zeros((n, 32), dtype=np.uint8)
            rows.append(v)
            arrs.append(a)
        d_s = torch.from_numpy(np.stack(arrs).copy()).to(dev)
        d_out = torch.zeros((batch, 96), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.g1_msm_device_batch_async(table, d_s.data_ptr(), n, batch, d_out.data_ptr())
        got = eng.g1_batch_to_affine_device(d_out.data_ptr(), batch)
        for q in range(batch):
            single = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s[q].data_ptr(), n))
            assert got[64 * q:64 * q + 64] == single, q
            t = sum(k * s for k, s in zip(ks, rows[q])) % O.R
            assert single == O.aff_to_bytes(O.scalar_mul(t, O.G1)), q
    finally:
        eng.msm_configure_glv(0)
        eng.bases_free(table)


def test_msm_2p22_identity(eng):
    """BASELINE.json configs[4] size (2^22 points per MSM): MSM([k_i G], [s_i]) = (sum k_i s_i) G, both recodings."""
    n = 1 << 22
    dev = torch.device("cuda", 0)
    ks, k_np = _workload(n, 21)
    ss, s_np = _workload(n, 22)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    d_s = torch.from_numpy(s_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        t = sum(k * s for k, s in zip(ks, ss)) % O.R
        want = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1), O.fe_to_bytes(t)))
        for glv in (-1, 1):
            eng.msm_configure_glv(glv)
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want
    finally:
        eng.msm_configure_glv(0)
        eng.bases_free(table)


def test_host_buffer_msm_sliced_pipeline(eng):
    """h2agg_g1_msm cuts inputs of >= 2^20 points into slices copied on a side stream under the previous slice's
    compute; ragged size, pageable and page-locked sources, against the resident-table path."""
    import ctypes
    n = (1 << 20) + 4099
    dev = torch.device("cuda", 0)
    ks, k_np = _workload(n, 31)
    ss, s_np = _workload(n, 32)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    d_s = torch.from_numpy(s_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        want = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
        t = sum(k * s for k, s in zip(ks, ss)) % O.R
        assert want == eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1), O.fe_to_bytes(t)))
        bases, scalars = eng.bases_download(table, 0, n), bytes(s_np.tobytes())
        assert eng.g1_batch_to_affine(eng.g1_msm(bases, scalars)) == want
        pb, ps = eng.host_alloc(64 * n), eng.host_alloc(32 * n)
        try:
            ctypes.memmove(pb, bases, 64 * n)
            ctypes.memmove(ps, scalars, 32 * n)
            for _ in range(2):
                assert eng.g1_batch_to_affine(eng.g1_msm(pb, ps, n)) == want
        finally:
            eng.host_free(pb)
            eng.host_free(ps)
    finally:
        eng.bases_free(table)


def test_throughput_mode_paths(eng):
    """Overlap (throughput) mode switches a large MSM to plain recoding and 32-bucket segments reduced by the 4-lane
    kernels, with tails on the side streams: several MSMs queued back to back, and a 16-MSM batch, must give the same
    points as the default mode."""
    n = 1 << 20
    dev = torch.device("cuda", 0)
    ks, k_np = _workload(n, 41)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        scal = [_workload(n, 50 + i) for i in range(3)]
        d_s = [torch.from_numpy(a.copy()).to(dev) for _v, a in scal]
        want = [eng.g1_batch_to_affine(eng.g1_msm_device(table, d.data_ptr(), n)) for d in d_s]
        t0 = sum(k * s for k, s in zip(ks, scal[0][0])) % O.R
        assert want[0] == O.aff_to_bytes(O.scalar_mul(t0, O.G1))
        d_out = torch.zeros((3, 96), dtype=torch.uint8, device=dev)
        eng.msm_set_tail_overlap(2)
        try:
            for rep in range(2):
                for i, d in enumerate(d_s):
                    eng.g1_msm_device_async(table, d.data_ptr(), n, d_out[i].data_ptr())
                eng.synchronize()
                assert eng.g1_batch_to_affine_device(d_out.data_ptr(), 3) == b"".join(want)
            m, B = (1 << 17) - 6, 16
            d_b = torch.from_numpy(np.stack([_workload(m, 70 + q)[1] for q in range(B)]).copy()).to(dev)
            d_bo = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
            eng.g1_msm_device_batch_async(table, d_b.data_ptr(), m, B, d_bo.data_ptr())
            got = eng.g1_batch_to_affine_device(d_bo.data_ptr(), B)
        finally:
            eng.msm_set_tail_overlap(0)
        for q in range(B):
            assert got[64 * q:64 * q + 64] == eng.g1_batch_to_affine(eng.g1_msm_device(table, d_b[q].data_ptr(), m)), q
    finally:
        eng.bases_free(table)


@pytest.mark.parametrize("n,cw", [(5000, 0), (5000, 11), (1 << 15, 0), (300, 20), ((1 << 17) - 6, 0), (70001, 20), ((1 << 19) - 6, 0),
                                  ((1 << 20) + 3, 0), ((1 << 22) - 6, 0)])
def test_fixed_base_levels(eng, n, cw):
    """h2agg_bases_precompute: MSMs over a table with fixed-base levels (single bucket set, digits looked up at level
    w) give the same points as the ordinary path — single, prefix, batched; zero / r-1 / one scalars included."""
    dev = torch.device("cuda", 0)
    ks, k_np = _workload(n, 81)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        B = 3
        rows = [_workload(n, 90 + q) for q in range(B)]
        arr = np.stack([a for _v, a in rows]).copy()
        arr[0, :7] = 0
        arr[0, 7] = np.frombuffer((O.R - 1).to_bytes(32, "little"), dtype=np.uint8)
        arr[0, 8] = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
        d_s = torch.from_numpy(arr).to(dev)
        m = max(1, n - n // 3)
        want_full = [eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s[q].data_ptr(), n)) for q in range(B)]
        want_prefix = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s[1].data_ptr(), m))
        t1 = sum(k * s for k, s in zip(ks, rows[1][0])) % O.R
        assert want_full[1] == O.aff_to_bytes(O.scalar_mul(t1, O.G1))
        eng.bases_precompute(table, cw)
        for q in range(B):
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s[q].data_ptr(), n)) == want_full[q], q
        assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s[1].data_ptr(), m)) == want_prefix
        d_out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
        eng.g1_msm_device_batch_async(table, d_s.data_ptr(), n, B, d_out.data_ptr())
        assert eng.g1_batch_to_affine_device(d_out.data_ptr(), B) == b"".join(want_full)
        assert eng.g1_batch_to_affine(eng.g1_msm_preloaded(table, bytes(arr[2].tobytes()))) == want_full[2]
    finally:
        eng.bases_free(table)


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "small", "top_digit_only", "half_zero", "u64", "u128", "sixty_long_partitions",
                                  "a_little_over_the_stage"])
def test_fixed_base_levels_c20_skewed_scalars(eng, kind):
    """the (level, point) sort of the big-table levels (csrc/fb_sort_kernels.hpp) under scalars that are not uniform: every key
    of a level in ONE bucket (level 2's over-long-partition path with wave-aggregated counters, level 1's one-counter tiles,
    the accumulation's over-long-bucket chunks), only the top digit's own slots in use, half the digits zero — against the
    ordinary path over the same table."""
    dev = torch.device("cuda", 0)
    n = (1 << 21) + 9 if kind == "a_little_over_the_stage" else (1 << 18) + 5
    ks, k_np = _workload(n, 181)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        rng = np.random.Generator(np.random.PCG64(77))
        if kind == "all_equal":
            vals = [0x1234567 << 100 | 0xabcdef] * n
        elif kind == "two_values":
            vals = [(O.R - 1) if i & 1 else (1 << 240) + 12345 for i in range(n)]
        elif kind == "small":
            vals = [int(v) for v in rng.integers(0, 1000, size=n)]
        elif kind == "top_digit_only":
            vals = [int(v) << 240 for v in rng.integers(1, 1 << 13, size=n)]
        elif kind == "u64":        # what instance columns hold: digit 3 has four bits -> its keys fill sixteen buckets of partition 0,
            vals = [int.from_bytes(rng.bytes(8), "little") for _ in range(n)]      # split over workgroups (k_fb_long_count / _place)
        elif kind == "u128":
            vals = [int.from_bytes(rng.bytes(16), "little") for _ in range(n)]
        elif kind == "sixty_long_partitions":
            # twelve equal digits per scalar, drawn from 90 partitions: 12 n / 90 = 35 k keys each in runs of ~290 — more very
            # long partitions than the list holds (64): the others go one workgroup each, tile-major
            vals = [sum(((int(v) % 90) * 256 + 1 + ((int(v) >> 8) % 7)) << (20 * w) for w in range(12)) for v in rng.integers(0, 1 << 30, size=n)]
        elif kind == "a_little_over_the_stage":
            # (n = 2^21 + 9) twelve equal digits from 768 partitions: 32 k keys each — over the level-2 stage's 28 k — in runs of
            # 32 per tile: the key-major shape of k_fb_bucket_sort_long
            vals = [sum(((int(v) % 768) * 256 + 1 + ((int(v) >> 10) % 256)) << (20 * w) for w in range(12)) for v in rng.integers(0, 1 << 30, size=n)]
        else:
            vals = [0 if i % 2 else int.from_bytes(rng.bytes(31), "little") for i in range(n)]
        arr = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(n, 32).copy()
        d_s = torch.from_numpy(arr).to(dev)
        want = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
        tot = sum(k * s for k, s in zip(ks, vals)) % O.R
        assert want == O.aff_to_bytes(O.scalar_mul(tot, O.G1))
        eng.bases_precompute(table, 20)
        assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want
        eng.msm_configure_lanes_per_bucket(2)            # every slot's run on two lanes + the slice combine, then the fold
        try:
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want
        finally:
            eng.msm_configure_lanes_per_bucket(0)
        assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n - 70001)) == \
            O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks[:n - 70001], vals)) % O.R, O.G1))
    finally:
        eng.bases_free(table)


@pytest.mark.parametrize("bad", [1 << 254, 1 << 255, (1 << 256) - 1, None])
def test_fixed_base_levels_c20_refuse_a_scalar_above_the_modulus(eng, pkg, bad):
    """a scalar >= r in a caller's device buffer over a table with c = 20 levels: H2AGG_ERR_NONCANONICAL, cleanly — with bits
    254 / 255 set its top digit names a partition past the level-1 sort's counters (LDS out of bounds, then gathers
    outside the level table); None: r itself, the smallest one.  The context and the table work afterwards."""
    dev = torch.device("cuda", 0)
    n = (1 << 18) + 77
    ks, k_np = _workload(n, 201)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        vals, s_np = _workload(n, 202)
        eng.bases_precompute(table, 20)
        good = torch.from_numpy(s_np.copy()).to(dev)
        want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, vals)) % O.R, O.G1))
        assert eng.g1_batch_to_affine(eng.g1_msm_device(table, good.data_ptr(), n)) == want
        arr = s_np.copy()
        for at in (0, 2047, 2048, n - 1, n // 2):            # first / last of a tile, the ragged last tile
            arr[at] = np.frombuffer((O.R if bad is None else bad).to_bytes(32, "little"), dtype=np.uint8)
        d_s = torch.from_numpy(arr).to(dev)
        for _ in range(3):
            with pytest.raises(pkg.H2AggError) as ei:
                eng.g1_msm_device(table, d_s.data_ptr(), n)
            assert ei.value.code == pkg.ERR_NONCANONICAL
        assert eng.g1_batch_to_affine(eng.g1_msm_device(table, good.data_ptr(), n)) == want
    finally:
        eng.bases_free(table)


def test_fixed_base_levels_refusals(eng, pkg):
    """h2agg_bases_precompute: an explicit width whose levels outgrow the packed sort item is refused unless it is 20 (the
    (level, point) sort's width); debug key pre_big lifts that for A/B runs through the two-array sort"""
    dev = torch.device("cuda", 0)
    n = (1 << 19) + 3
    _ks, k_np = _workload(n, 191)
    d_k = torch.from_numpy(k_np.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    try:
        with pytest.raises(pkg.H2AggError) as ei:
            eng.bases_precompute(table, 16)
        assert ei.value.code == pkg.ERR_INVALID
        _v, s_np = _workload(n, 192)
        d_s = torch.from_numpy(s_np.copy()).to(dev)
        want = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
        eng.bases_precompute(table, 0)                       # auto: c = 20 here
        assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want
        eng.debug_configure("pre_big", 1)
        try:
            eng.bases_precompute(table, 19)                  # 14 levels through the two-array sort
            assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want
        finally:
            eng.debug_configure("pre_big", 0)
    finally:
        eng.bases_free(table)
