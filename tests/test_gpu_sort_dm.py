"""GPU: the digit-major bucket sort (csrc/sort_kernels.hpp: k_dm_digits / k_dm_partition / k_dm_bucket_sort) that plain
16-bit-window MSMs of 2^16 .. 2^22 points take.  Expected values are (sum k_i s_i) * G from the Python oracle; every case
is also run through the packed two-level sort (h2agg_msm_configure_sort(ctx, 0, -3)) and must give the same bytes."""
import numpy as np
import pytest
import torch

from oracle import bn254 as O

pytestmark = pytest.mark.gpu

R = O.R


def _bytes(vals):
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(len(vals), 32)


def _rand(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.bytes(64 * n)
    return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % R for i in range(n)]


def _run(eng, ks, ss):
    n = len(ks)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(_bytes(ks).copy()).to(dev)
    d_s = torch.from_numpy(_bytes(ss).copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % R, O.G1))
    try:
        eng.msm_configure(16, 0, 0)
        for glv in (-1, 1):   # 16 windows of n keys / 8 windows of 2n keys (both halves of the endomorphism split)
            eng.msm_configure_glv(glv)
            got = {}
            for name, tile in (("digit-major", 0), ("packed", -3)):
                eng.msm_configure_sort(0, tile)
                got[name] = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
            assert got["digit-major"] == want, glv
            assert got["packed"] == want, glv
    finally:
        eng.msm_configure_sort(0, 0)
        eng.msm_configure(0, 0, 0)
        eng.msm_configure_glv(0)
        eng.bases_free(table)


@pytest.mark.parametrize("n", [1 << 16, (1 << 16) + 5, 100003, (1 << 18) + 8191, 1 << 20, (1 << 21) + 12345, 1 << 22])
def test_random_scalars(eng, n):
    _run(eng, _rand(n, 7 * n + 1), _rand(n, 7 * n + 2))


def test_ragged_sizes_around_the_tile(eng):
    for n in (8192 * 9 - 1, 8192 * 9 + 1, 8192 * 8 + 4097):
        _run(eng, _rand(n, n), _rand(n, n + 1))


def test_structured_scalars(eng):
    """zero digits, digits at both ends of the signed range, carries that ripple through every window, one value in every
    lane (a single bucket per window holds everything: the over-long partition path), r - 1, 0"""
    n = (1 << 16) + 77
    ks = _rand(n, 5)
    half = sum(0x8000 << (16 * w) for w in range(15))          # every digit exactly +2^15
    ripple = sum(0xffff << (16 * w) for w in range(15))        # ... 0xffff: digit -1 then carries all the way up
    over = sum(0x8001 << (16 * w) for w in range(15))          # first negative magnitude (2^15 - 1)
    pats = [0, 1, R - 1, half, ripple, over, 1 << 240, (1 << 253) + 12345, 0x10000, 0xffff0000ffff]
    _run(eng, ks, [pats[i % len(pats)] for i in range(n)])
    _run(eng, ks, [R - 1] * n)
    _run(eng, ks, [0x1234_5678_9abc_def0_1111_2222_3333_4444_5555_6666_7777_8888_9999_aaaa_bbbb % R] * n)
    _run(eng, ks, [0] * n)
    # endomorphism split: scalars k1 + lambda * k2 with halves full of 0x8000 / 0x7fff / 0xffff digits, both signs
    lam = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
    halves = [sum(0x8000 << (16 * w) for w in range(7)), sum(0x7fff << (16 * w) for w in range(7)),
              sum(0xffff << (16 * w) for w in range(7)), (1 << 112) - 1, 1 << 111, 0x8000, 0x7fff, 0x18000]
    glv = []
    for a in halves:
        for b in halves:
            for sa in (1, -1):
                for sb in (1, -1):
                    glv.append((sa * a + sb * b * lam) % R)
    _run(eng, ks, [glv[i % len(glv)] for i in range(n)])


def test_skewed_partitions(eng):
    """three quarters of the scalars share their low digits (one level-2 partition far beyond the LDS stage), the rest random"""
    n = 1 << 17
    ks = _rand(n, 9)
    rnd = _rand(n, 10)
    ss = [(rnd[i] & ~0xffffffff) | 0x12345678 if i % 4 else rnd[i] for i in range(n)]
    _run(eng, ks, [s % R for s in ss])


@pytest.mark.parametrize("frac", [1.0, 0.5, 0.05])
def test_equal_scalars(eng, frac):
    """all / half / a twentieth of the scalars equal: one bucket per window holds that share of the row — the skew pass of
    level 2 (dm_long_pass: tile-major, wave-aggregated slots) and the over-long-bucket kernels; under GLV the two halves
    of the split put their runs in opposite halves of the row"""
    n = (1 << 18) + 333
    ks, rnd = _rand(n, 31), _rand(n, 32)
    m = int(n * frac)
    _run(eng, ks, [rnd[0] if i < m else rnd[i] for i in range(n)])


def test_moderately_long_partitions(eng):
    """digits confined to a sixteenth of the range in every window: partitions two to eight stages long with spread
    buckets (the path the top window's partitions take on random scalars)"""
    n = 1 << 18
    ks, rnd = _rand(n, 33), _rand(n, 34)
    mask = sum(0x0fff << (16 * w) for w in range(16))
    _run(eng, ks, [(r & mask) % R for r in rnd])


def test_reduction_paths_agree(eng):
    """16-bit windows take the two-dimensional bucket reduction (k_msm_reduce2d_*); an explicit reduce_segment keeps the
    segment kernels: same bytes, with and without GLV (8 / 16 windows), one MSM after another on rotating tail slots"""
    n = (1 << 16) + 123
    ks, ss = _rand(n, 21), _rand(n, 22)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(_bytes(ks).copy()).to(dev)
    d_s = torch.from_numpy(_bytes(ss).copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % R, O.G1))
    try:
        for glv in (-1, 1):
            eng.msm_configure_glv(glv)
            for seg in (0, 32, 0, 8, 0):
                eng.msm_configure(16, seg, 0)
                assert eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n)) == want, (glv, seg)
    finally:
        eng.msm_configure(0, 0, 0)
        eng.msm_configure_glv(0)
        eng.bases_free(table)


# ---- 17-bit windows (15 windows of 2^16 buckets; 32-bit digit codes, 256 x 256 reduction grid) ---------------------------
def _run17(eng, ks, ss):
    n = len(ks)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(_bytes(ks).copy()).to(dev)
    d_s = torch.from_numpy(_bytes(ss).copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % R, O.G1))
    try:
        eng.msm_configure_glv(-1)
        got = {}
        for c in (17, 16):
            eng.msm_configure(c, 0, 0)
            got[c] = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
        eng.msm_configure(17, 0, 0)
        eng.msm_configure_sort(0, -3)       # the packed two-level sort + segment reduction: the same plan through other kernels
        got["packed17"] = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
        assert got[17] == want
        assert got[16] == want and got["packed17"] == want
    finally:
        eng.msm_configure_sort(0, 0)
        eng.msm_configure(0, 0, 0)
        eng.msm_configure_glv(0)
        eng.bases_free(table)


@pytest.mark.parametrize("n", [1 << 16, (1 << 16) + 5, 100003, (1 << 18) + 8191, 1 << 20, (1 << 21) + 12345])
def test_c17_random_scalars(eng, n):
    _run17(eng, _rand(n, 11 * n + 1), _rand(n, 11 * n + 2))


def test_c17_structured_scalars(eng):
    """17-bit digits at both ends of the signed range (+2^16, -(2^16 - 1)), carries rippling through all 15 windows, the top
    window's 16 bits + carry, zero digits, one value in every lane (one bucket per window holds everything)"""
    n = (1 << 16) + 77
    ks = _rand(n, 6)
    half = sum(0x10000 << (17 * w) for w in range(14))         # every digit exactly +2^16
    ripple = sum(0x1ffff << (17 * w) for w in range(14))       # ... all ones: digit -1 then carries all the way up
    over = sum(0x10001 << (17 * w) for w in range(14))         # first negative magnitude (2^16 - 1)
    pats = [0, 1, R - 1, half, ripple, over, 1 << 238, (1 << 253) + 12345, 0x20000, 0x1ffff0001ffff, (1 << 254) - 1 - (1 << 200)]
    pats = [v % R for v in pats]
    _run17(eng, ks, [pats[i % len(pats)] for i in range(n)])
    _run17(eng, ks, [R - 1] * n)
    _run17(eng, ks, [0] * n)
    _run17(eng, ks, [0x1234_5678_9abc_def0_1111_2222_3333_4444_5555_6666_7777_8888_9999_aaaa_bbbb % R] * n)


def test_large_plain_msms_take_17_bit_windows_by_themselves(eng):
    """choose_window: from 1.5 * 2^20 points on a plain (non-GLV) MSM runs on 15 windows of 17 bits without being told to — the
    plan of back-to-back MSMs in overlap mode; the explicit 16-bit plan must give the same point"""
    n = (3 << 19) + 11
    ks, ss = _rand(n, 9101), _rand(n, 9102)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(_bytes(ks).copy()).to(dev)
    d_s = torch.from_numpy(_bytes(ss).copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % R, O.G1))
    out = torch.zeros(96 * 3, dtype=torch.uint8, device=dev)
    try:
        eng.msm_set_tail_overlap(2)
        for i in range(3):
            eng.g1_msm_device_async(table, d_s.data_ptr(), n, out.data_ptr() + 96 * i)
        eng.synchronize()
        res = bytes(out.cpu().numpy().tobytes())
        for i in range(3):
            assert eng.g1_batch_to_affine(res[96 * i:96 * i + 96]) == want
        eng.msm_configure(16, 0, 0)
        eng.g1_msm_device_async(table, d_s.data_ptr(), n, out.data_ptr())
        eng.synchronize()
        assert eng.g1_batch_to_affine(bytes(out[:96].cpu().numpy().tobytes())) == want
    finally:
        eng.msm_configure(0, 0, 0)
        eng.msm_set_tail_overlap(0)
        eng.bases_free(table)
