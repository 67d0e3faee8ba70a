"""GPU: slices of ONE multi_exp that share a bucket set (`msm_run`'s chain modes in csrc/h2agg.hip): the host-buffer MSM
(`h2agg_g1_msm`, the call behind ArithEccChip::multi_exp in the drop-in, mock/arith/ecc.rs:106-129) cuts its input into slices
that cross PCIe while the previous one is accumulated; only the last slice is followed by the bucket reduction / Horner tail.
Every slice resumes the bucket sums the earlier ones left — including buckets that are over-long (chunked path) in some slices
and ordinary in others, and the lanes-per-bucket slice slots of small plans.

Expected values: (sum k_i s_i) G from the Python oracle; the unchained scheme (H2AGG_PCIE_CHAIN=0) must agree bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import bn254 as O
from tests.util import fr_bytes

pytestmark = pytest.mark.gpu


def _vals(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.bytes(64 * n)
    return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % O.R for i in range(n)]


def _bases(eng, ks):
    """k_i * G computed on the device, returned as host bytes (affine, 64 B each)"""
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


_HOOK = {"H2AGG_PCIE_SLICES": ("pcie_slices", 0), "H2AGG_PCIE_GLV": ("pcie_glv", 0), "H2AGG_PCIE_CHAIN": ("pcie_chain", 1),
         "H2AGG_COMB_MSM": ("comb_msm", 1)}
_ENG = []   # the engine the hooks act on (set by the autouse fixture below)


class _Env:
    """per-call test hooks of the library (h2agg_debug_configure; they used to be environment variables, hence the names)"""

    def __init__(self, **kv):
        self.kv = {k: int(v) for k, v in kv.items()}

    def __enter__(self):
        for k, v in self.kv.items():
            _ENG[0].debug_configure(_HOOK[k][0], v)

    def __exit__(self, *a):
        for k in self.kv:
            _ENG[0].debug_configure(*_HOOK[k])


@pytest.fixture(autouse=True)
def _hooks_engine(eng):
    _ENG[:] = [eng]
    yield
    for key, default in _HOOK.values():
        eng.debug_configure(key, default)


def _scalars(kind, n, seed):
    ss = _vals(n, seed)
    q = n // 4
    if kind == "equal":
        ss = [ss[0]] * n                         # one bucket per window, over-long in every slice
    elif kind == "first_quarter_equal":
        ss[:q] = [ss[0]] * q                     # over-long in slice 0 only: later slices resume it as an ordinary bucket
    elif kind == "last_quarter_equal":
        ss[n - q:] = [ss[-1]] * q                # ordinary partial sums first, the chunked path on top in the last slice
    elif kind == "small":
        ss = [s % 1000 for s in ss]              # upper windows stay empty in every slice
    elif kind == "sparse":
        ss = [s if i % 7 == 0 else 0 for i, s in enumerate(ss)]   # most keys dropped; some slices leave most buckets untouched
    return ss


@pytest.mark.parametrize("glv", [1, -1])
@pytest.mark.parametrize("kind", ["random", "equal", "first_quarter_equal", "last_quarter_equal", "small", "sparse"])
def test_chained_slices_resume_bucket_sums(eng, kind, glv):
    n = (1 << 14) + 37                           # ragged last slice
    ks = _vals(n, 7001)
    ss = _scalars(kind, n, 7002)
    if kind == "random":
        ks[5] = ks[5000]                     # the same base with the same scalar in two different slices: P + P on resume
        ss[5] = ss[5000]
        ks[9] = (-ks[8200]) % O.R                # a base in one slice, its negation in another, same scalar: the bucket returns to the identity
        ss[9] = ss[8200]
    bases, sb, want = _bases(eng, ks), fr_bytes(ss), _want(ks, ss)
    with _Env(H2AGG_PCIE_SLICES=4, H2AGG_PCIE_GLV=glv):
        got = eng.g1_batch_to_affine(eng.g1_msm(bases, sb))
        assert got == want
        with _Env(H2AGG_PCIE_CHAIN=0):
            assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want


@pytest.mark.parametrize("glv", [1, -1])
@pytest.mark.parametrize("c,big", [(8, 64), (5, 0), (13, 0)])
def test_chained_slices_small_plans_with_slice_slots(eng, c, big, glv):
    """narrow windows: few buckets, several lanes per bucket (slice slots resumed one by one, folded once at the end),
    and a low over-long threshold so that the chunked path and the slots meet"""
    n = 6000
    ks = _vals(n, 7101 + c)
    ss = _scalars("first_quarter_equal", n, 7102 + c)
    bases, sb, want = _bases(eng, ks), fr_bytes(ss), _want(ks, ss)
    eng.msm_configure(window_bits=c, big_bucket_threshold=big)
    eng.msm_configure_glv(glv)
    try:
        for slices in (2, 3, 5):
            with _Env(H2AGG_PCIE_SLICES=slices):
                assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want, slices
    finally:
        eng.msm_configure()
        eng.msm_configure_glv(0)


@pytest.mark.parametrize("slices,glv", [(2, 1), (4, -1), (8, 1), (16, -1)])
def test_chained_slices_digit_major_sizes(eng, slices, glv):
    """2^20 + 5 points: 16-bit windows, digit-major sort and two-dimensional reduction in every slice"""
    n = (1 << 20) + 5
    ks = _vals(n, 7201)
    ss = _vals(n, 7202)
    bases, sb, want = _bases(eng, ks), fr_bytes(ss), _want(ks, ss)
    with _Env(H2AGG_PCIE_SLICES=slices, H2AGG_PCIE_GLV=glv):
        assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want
    # the default cut (no knobs) as well
    assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want


def test_chain_leaves_the_context_clean(eng):
    """an ordinary MSM right after a chained one (the slot rotation and the slice slots must not leak)"""
    n = 1 << 13
    ks, ss = _vals(n, 7301), _vals(n, 7302)
    bases, sb, want = _bases(eng, ks), fr_bytes(ss), _want(ks, ss)
    with _Env(H2AGG_PCIE_SLICES=4):
        assert eng.g1_batch_to_affine(eng.g1_msm(bases, sb)) == want
    m = 3000
    assert eng.g1_batch_to_affine(eng.g1_msm(bases[:64 * m], sb[:32 * m])) == _want(ks[:m], ss[:m])


def test_chained_slices_random_shapes_agree_with_the_resident_msm(eng):
    """random sizes / slice counts / plans / scalar patterns: the chained host-buffer MSM against the resident one over the same
    table (whose own parity with the oracle is tests/test_gpu_parity.py's and tests/test_gpu_scale.py's business)"""
    rng = np.random.Generator(np.random.PCG64(7401))
    n_max = 300_000
    ks = _vals(n_max, 7402)
    arr = np.frombuffer(fr_bytes(ks), dtype=np.uint8).reshape(n_max, 32)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(arr.copy()).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n_max)
    bases = eng.bases_download(table, 0, n_max)
    try:
        for case in range(10):
            n = int(rng.integers(3000, n_max))
            slices = int(rng.integers(2, 10))
            glv = int(rng.choice([1, -1]))
            kind = str(rng.choice(["random", "first_quarter_equal", "last_quarter_equal", "sparse", "small"]))
            ss = _scalars(kind, n, 7500 + case)
            sb = fr_bytes(ss)
            d_s = torch.from_numpy(np.frombuffer(sb, dtype=np.uint8).copy()).to(dev)
            want = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
            with _Env(H2AGG_PCIE_SLICES=slices, H2AGG_PCIE_GLV=glv):
                got = eng.g1_batch_to_affine(eng.g1_msm(bases[:64 * n], sb))
            assert got == want, (case, n, slices, glv, kind)
    finally:
        eng.bases_free(table)


# ---- small multi_exps over the leading bases of a fixed table: the table's comb (csrc/scalar_mul_kernels.hpp k_comb_msm) -----
def test_small_msms_over_a_fixed_table_take_its_comb(eng, pkg):
    """assign_instance_commitment with a handful of public inputs (verify.rs:574-649): sum_i v_i * g_lagrange[i].  A table with
    fixed-base levels (h2agg_bases_precompute) answers MSMs of <= 256 scalars from a comb of its leading bases — no buckets, no
    doubling chain.  Against the reference loop (scalar multiplication + addition per term), for every length class, with
    zero / all-ones / maximal scalars and identity bases; the bucket path (H2AGG_COMB_MSM=0, and n = 257) must agree."""
    n_table = 300
    ks = _vals(n_table, 7601)
    ks[3] = 0                                    # an identity base inside the comb's range
    ks[255] = ks[254]                            # equal bases
    bases = _bases(eng, ks)
    h = eng.bases_upload(bases)
    dev = torch.device("cuda", 0)
    try:
        eng.bases_precompute(h)
        pats = [0, 1, O.R - 1, (1 << 248) - 1, 0xff, 0x0100, int.from_bytes(b"\x01" * 31 + b"\x00", "little"), 1 << 253]
        for n in (1, 2, 7, 64, 255, 256, 257):
            batch = 5
            cols = []
            for q in range(batch):
                ss = _vals(n, 7700 + 10 * n + q)
                for j, p in enumerate(pats):
                    if q == 0 and j < n:
                        ss[j] = p
                if q == 1:
                    ss = [0] * n                 # an all-zero column: the identity
                cols.append(ss)
            flat = b"".join(fr_bytes(ss) for ss in cols)
            d_s = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).to(dev)
            out = torch.zeros(96 * batch, dtype=torch.uint8, device=dev)
            want = [O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % O.R, O.G1)) for ss in cols]
            for env in ({}, {"H2AGG_COMB_MSM": "0"}):
                with _Env(**env):
                    eng.g1_msm_device_batch_async(h, d_s.data_ptr(), n, batch, out.data_ptr())
                    eng.synchronize()
                    got = bytes(out.cpu().numpy().tobytes())
                    for q in range(batch):
                        assert eng.g1_batch_to_affine(got[96 * q:96 * q + 96]) == want[q], (n, q, env)
                    assert eng.g1_batch_to_affine(eng.g1_msm_device(h, d_s.data_ptr(), n)) == want[0], (n, env)
        # a scalar >= r is refused on the comb route as on the bucket route
        bad = (O.R + 3).to_bytes(32, "little") + bytes(32)
        d_bad = torch.from_numpy(np.frombuffer(bad, dtype=np.uint8).copy()).to(dev)
        with pytest.raises(pkg.H2AggError) as ei:
            eng.g1_msm_device(h, d_bad.data_ptr(), 2)
        assert ei.value.code == pkg.ERR_NONCANONICAL
    finally:
        eng.bases_free(h)


def test_chained_slices_over_projective_points(eng):
    """h2agg_g1_msm_jac (the trait hands over Vec<C::CurveExt>: x || y || z, normalised on the device slice by slice)"""
    from tests.util import to_jac_bytes
    n = 5000
    ks, ss = _vals(n, 7801), _vals(n, 7802)
    ks[17] = 0                                   # the identity among the points (z = 0)
    aff = _bases(eng, ks)
    zs = [(z % O.P) or 1 for z in _vals(n, 7803)]
    jac = bytearray(to_jac_bytes(aff, zs))
    jac[96 * 17:96 * 18] = bytes(32) + (1).to_bytes(32, "little") + bytes(32)
    want = _want(ks, ss)
    for slices in (1, 3, 4):
        with _Env(H2AGG_PCIE_SLICES=slices):
            assert eng.g1_batch_to_affine(eng.g1_msm_jac(bytes(jac), fr_bytes(ss))) == want, slices


@pytest.mark.parametrize("fill", ["random", "ones"])
def test_non_canonical_scalars_are_refused_on_every_device_route(eng, pkg, fill):
    """random 256-bit values / all ones as scalars (what a buffer that was never filled looks like): every resident-table route
    — fixed-base batches, the packed and the digit-major sort, GLV or not, overlap on or off — must end in ERR_NONCANONICAL,
    never in a fault (an input that CHANGES during the call is another matter: include/h2agg.h)"""
    dev = torch.device("cuda", 0)
    rng = np.random.Generator(np.random.PCG64(7901))
    k = rng.integers(0, 256, size=(1 << 17, 32), dtype=np.uint8)
    k[:, 31] &= 0x1F
    d_k = torch.from_numpy(k.copy()).to(dev)
    torch.cuda.synchronize()
    t_small = eng.bases_generate(d_k.data_ptr(), 4096)
    t_big = eng.bases_generate(d_k.data_ptr(), 1 << 17)
    eng.bases_precompute(t_small)

    def garbage(shape):
        if fill == "ones":
            g = torch.full(shape, 255, dtype=torch.uint8, device=dev)
        else:
            g = torch.from_numpy(rng.integers(0, 256, size=shape, dtype=np.uint8)).to(dev)
            g[..., 31] |= 0x80                     # (every value >= 2^255 > r)
        torch.cuda.synchronize()
        return g

    def refused(fn):
        with pytest.raises(pkg.H2AggError) as ei:
            fn()
            eng.synchronize()
        assert ei.value.code == pkg.ERR_NONCANONICAL

    try:
        for ovl in (0, 2):
            eng.msm_set_tail_overlap(ovl)
            for batch in (2, 8):
                g, o = garbage((batch, 4090, 32)), torch.zeros((batch, 96), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                refused(lambda: eng.g1_msm_device_batch_async(t_small, g.data_ptr(), 4090, batch, o.data_ptr()))
            for n in (5, 200, 4090, 1 << 16, 1 << 17):
                g, o = garbage((n, 32)), torch.zeros(96, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                for glv in (1, -1):
                    eng.msm_configure_glv(glv)
                    refused(lambda: eng.g1_msm_device_async(t_big, g.data_ptr(), n, o.data_ptr()))
                eng.msm_configure_glv(0)
        # the context still works
        good = torch.from_numpy(k[:1000].copy()).to(dev)
        torch.cuda.synchronize()
        eng.g1_msm_device(t_big, good.data_ptr(), 1000)
    finally:
        eng.msm_configure_glv(0)
        eng.msm_set_tail_overlap(0)
        eng.bases_free(t_small)
        eng.bases_free(t_big)
