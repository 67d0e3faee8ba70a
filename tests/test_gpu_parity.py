"""GPU parity tests: every C-ABI entry point against the oracle on the same seeded inputs (bit-exact).

Mirrors the reference's own checks for the pure-calculation context: the algebraic identities of
halo2-ecc-circuit-lib/src/tests/five_native_ecc.rs:60-240 (add / mul / shamir incl. zero scalar and
identity point) and the Mock-chip runs of halo2-snark-aggregator-api/src/tests/systems/halo2/*/.
"""
import pytest

from oracle import bn254 as O, cref
from tests.util import G_BYTES, fr_bytes, norm, points_from_scalars, rand_frs, to_jac_bytes

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ Fr
@pytest.mark.parametrize("op", [0, 1, 2, 3, 4])
def test_fr_batch_op(eng, op):
    rng = O.SplitMix64(100 + op)
    n = 1000
    a = rand_frs(rng, n - 6) + [0, 1, O.R - 1, 2, O.R - 2, 1]
    b = rand_frs(rng, n - 6) + [0, O.R - 1, O.R - 1, 1, 3, 0]
    if op == 4:
        a = [x if x else 7 for x in a]
    ab, bb = fr_bytes(a), fr_bytes(b)
    got = eng.fr_batch_op(op, ab, bb if op <= 2 else None)
    assert got == cref.field_batch_op(0, op, ab, bb if op <= 2 else None, n)


def test_fr_div_and_pow_constant(eng, pkg):
    """MockFieldChip::div (mock/arith/field.rs:107-114) and ArithFieldChip::pow_constant (arith/field.rs:83-104)"""
    rng = O.SplitMix64(0xD1F)
    a = rand_frs(rng, 500) + [0, 1, O.R - 1]
    b = [x or 3 for x in rand_frs(rng, 500)] + [5, O.R - 1, 1]
    got = eng.fr_batch_op(5, fr_bytes(a), fr_bytes(b))
    assert got == fr_bytes([x * pow(y, -1, O.R) % O.R for x, y in zip(a, b)])
    with pytest.raises(pkg.DivisionByZero):
        eng.fr_batch_op(5, fr_bytes([1, 2]), fr_bytes([3, 0]))
    for e in (1, 2, 3, 7, 1 << 17, (1 << 32) + 5, (1 << 64) - 1):
        assert eng.fr_batch_pow_constant(fr_bytes(a), e) == fr_bytes([pow(x, e, O.R) for x in a])
    with pytest.raises(pkg.H2AggError):
        eng.fr_batch_pow_constant(fr_bytes(a), 0)


def test_fr_inv_zero_is_an_error(eng, pkg):
    # MockFieldChip::div: `b.invert().unwrap()` panics on zero (mock/arith/field.rs:113)
    with pytest.raises(pkg.DivisionByZero):
        eng.fr_batch_op(4, fr_bytes([5, 0, 9]))


def test_fr_noncanonical_rejected(eng, pkg):
    bad = (O.R).to_bytes(32, "little")
    with pytest.raises(pkg.H2AggError) as ei:
        eng.fr_batch_op(0, bad, fr_bytes([1]))
    assert ei.value.code == pkg.ERR_NONCANONICAL


def test_fr_empty_is_noop(eng):
    assert eng.fr_batch_op(2, b"", b"") == b""


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 1000, 5003])
def test_fr_mul_add_accumulate(eng, n):
    rng = O.SplitMix64(7 + n)
    v = rand_frs(rng, n)
    b = rng.fr()
    acc = 0
    for x in v:
        acc = (acc * b + x) % O.R                       # arith/field.rs:68-81
    assert eng.fr_mul_add_accumulate(fr_bytes(v), O.fe_to_bytes(b)) == O.fe_to_bytes(acc)
    assert cref.fr_mul_add_accumulate(fr_bytes(v), n, O.fe_to_bytes(b)) == O.fe_to_bytes(acc)


@pytest.mark.parametrize("n", [0, 1, 300, 1025])
def test_fr_sum_with_coeff_and_constant(eng, n):
    rng = O.SplitMix64(70 + n)
    x, cf, b = rand_frs(rng, n), rand_frs(rng, n), rng.fr()
    want = (b + sum(p * q for p, q in zip(x, cf))) % O.R   # mock/arith/field.rs:124-135
    assert eng.fr_sum_with_coeff_and_constant(fr_bytes(x), fr_bytes(cf), O.fe_to_bytes(b)) == O.fe_to_bytes(want)


# ------------------------------------------------------------------ G1 element-wise
def _edge_pairs():
    rng = O.SplitMix64(5)
    ks = rand_frs(rng, 8)
    pts = [O.scalar_mul(k, O.G1) for k in ks]
    P, Q = pts[0], pts[1]
    pairs = [(P, Q), (P, P), (P, O.neg(P)), (O.INF, P), (P, O.INF), (O.INF, O.INF), (O.G1, O.G1), (Q, O.neg(Q))]
    return pairs


@pytest.mark.parametrize("subtract", [False, True])
def test_g1_batch_add_edges(eng, subtract):
    pairs = _edge_pairs()
    rng = O.SplitMix64(11)
    a = b"".join(O.jac_to_bytes(p, rng.fr() % O.P or 1) for p, _ in pairs)
    b = b"".join(O.jac_to_bytes(q, rng.fr() % O.P or 1) for _, q in pairs)
    got = norm(eng, eng.g1_batch_add(a, b, subtract))
    want = b"".join(O.aff_to_bytes(O.sub(p, q) if subtract else O.add(p, q)) for p, q in pairs)
    assert got == want
    assert norm(None, cref.g1_batch_add(a, b, len(pairs), subtract)) == want


def test_g1_batch_add_random(eng):
    rng = O.SplitMix64(12)
    n = 700
    pa, pb = points_from_scalars(rand_frs(rng, n)), points_from_scalars(rand_frs(rng, n))
    za = [rng.fr() % O.P or 1 for _ in range(n)]
    zb = [1 if i % 3 == 0 else (rng.fr() % O.P or 1) for i in range(n)]
    a, b = to_jac_bytes(pa, za), to_jac_bytes(pb, zb)
    assert norm(eng, eng.g1_batch_add(a, b)) == norm(None, cref.g1_batch_add(a, b, n))


def test_g1_add_identity_five_native_ecc(eng):
    # five_native_ecc.rs:60-88: s1*G + s2*G == (s1+s2)*G
    rng = O.SplitMix64(13)
    s1, s2 = rng.fr(), rng.fr()
    j = eng.g1_batch_scalar_mul(G_BYTES * 3, fr_bytes([s1, s2, s1 + s2]))
    lhs = norm(eng, eng.g1_batch_add(j[:96], j[96:192]))
    assert lhs == norm(eng, j[192:288])


def test_g1_batch_scalar_mul(eng):
    rng = O.SplitMix64(14)
    n = 200
    ks = rand_frs(rng, n)
    bases = bytearray(points_from_scalars(ks))
    bases[64 * 5:64 * 6] = bytes(64)                         # identity base (five_native_ecc.rs:118-150)
    ss = rand_frs(rng, n)
    ss[0], ss[1], ss[2], ss[3] = 0, 1, O.R - 1, 2
    got = norm(eng, eng.g1_batch_scalar_mul(bytes(bases), fr_bytes(ss)))
    want = norm(None, cref.g1_batch_scalar_mul(bytes(bases), fr_bytes(ss), n))
    assert got == want
    assert got[:64] == bytes(64) and got[64 * 5:64 * 6] == bytes(64)


def test_g1_batch_to_affine(eng):
    rng = O.SplitMix64(15)
    n = 300
    aff = bytearray(points_from_scalars(rand_frs(rng, n)))
    aff[64 * 7:64 * 8] = bytes(64)
    jac = to_jac_bytes(bytes(aff), [rng.fr() % O.P or 1 for _ in range(n)])
    assert eng.g1_batch_to_affine(jac) == bytes(aff)


def test_inversion_edges_and_shapes(eng):
    """The device inverse is a safegcd (divsteps) implementation on signed 29-bit limbs (csrc/fp.hpp): exercise limb
    boundaries, near-modulus values and short / long operands in both fields against exact big-int inverses."""
    rng = O.SplitMix64(77)
    for mod, is_fr in ((O.R, True), (O.P, False)):
        xs = [1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, mod >> 1, (1 << 29) - 1, 1 << 29, (1 << 29) + 1,
              (1 << 58) - 1, 1 << 58, (1 << 232) + 1, (1 << 253) + 1, 1 << 128, (1 << 128) - 1]
        xs += [(rng.fr() * rng.fr()) % (1 << k) or 1 for k in range(1, 254, 2)]
        xs += [mod - ((rng.fr() % (1 << k)) or 1) for k in range(1, 250, 3)]
        xs += [rng.fr() % mod or 1 for _ in range(3000)]
        if is_fr:
            got = eng.fr_batch_op(4, fr_bytes(xs), None)
            want = b"".join(pow(x, -1, mod).to_bytes(32, "little") for x in xs)
            assert got == want
        else:   # Fq inverses are reached through to_affine: (X, Y, Z) = (x z^2, y z^3, z) must come back as (x, y)
            n = len(xs)
            aff = points_from_scalars(rand_frs(rng, 8)) * (n // 8 + 1)
            aff = aff[:64 * n]
            assert eng.g1_batch_to_affine(to_jac_bytes(aff, xs)) == aff


@pytest.mark.parametrize("n", [0, 1, 2, 8, 300])
def test_g1_sum(eng, n):
    rng = O.SplitMix64(16 + n)
    ks = rand_frs(rng, n)
    jac = to_jac_bytes(points_from_scalars(ks), [rng.fr() % O.P or 1 for _ in range(n)])
    want = O.aff_to_bytes(O.scalar_mul(sum(ks) % O.R, O.G1))
    assert norm(eng, eng.g1_sum(jac)) == want


# ------------------------------------------------------------------ multi_exp
def _msm_case(rng, n, kind):
    ks = rand_frs(rng, n)
    ss = rand_frs(rng, n)
    if kind == "edges" and n >= 8:
        ss[0] = 0
        ss[1] = O.R - 1
        ss[2] = 1
        ks[3] = ks[4]                 # duplicate base: P + P inside a bucket when scalars collide
        ss[3] = ss[4]
        ks[5] = (-ks[6]) % O.R        # a base and its negation with the same scalar: bucket sums to identity
        ss[5] = ss[6]
    if kind == "equal_scalars":
        ss = [ss[0]] * n              # every point in the same bucket of every window (big-bucket path)
    if kind == "small_scalars":
        ss = [s % 1000 for s in ss]   # upper windows empty
    bases = bytearray(points_from_scalars(ks))
    if kind == "edges" and n >= 8:
        bases[64 * 7:64 * 8] = bytes(64)   # identity base
        ks[7] = 0
    want_k = sum(k * s for k, s in zip(ks, ss)) % O.R
    return bytes(bases), fr_bytes(ss), O.aff_to_bytes(O.scalar_mul(want_k, O.G1))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 256, 1024])
@pytest.mark.parametrize("kind", ["random", "edges"])
def test_msm_vs_reference_algorithm(eng, n, kind):
    rng = O.SplitMix64(1000 + n)
    bases, sb, want = _msm_case(rng, n, kind)
    got = norm(eng, eng.g1_msm(bases, sb))
    assert got == cref.multi_exp_naive(bases, sb, n)      # mock/arith/ecc.rs:106-129 restated
    assert got == want                                    # (sum k_i s_i) G


@pytest.mark.parametrize("glv", [1, -1])
@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16])
def test_msm_every_window_size(eng, c, glv):
    rng = O.SplitMix64(2000 + c)
    n = 600
    bases, sb, want = _msm_case(rng, n, "edges")
    eng.msm_configure(window_bits=c)
    eng.msm_configure_glv(glv)
    try:
        assert norm(eng, eng.g1_msm(bases, sb)) == want
    finally:
        eng.msm_configure()
        eng.msm_configure_glv(0)


@pytest.mark.parametrize("glv", [1, -1])
@pytest.mark.parametrize("kind", ["equal_scalars", "small_scalars"])
def test_msm_skewed_buckets(eng, kind, glv):
    rng = O.SplitMix64(3000)
    n = 3000
    bases, sb, want = _msm_case(rng, n, kind)
    eng.msm_configure_glv(glv)
    eng.msm_configure(window_bits=8, big_bucket_threshold=64)
    try:
        assert norm(eng, eng.g1_msm(bases, sb)) == want
        eng.msm_configure()
        assert norm(eng, eng.g1_msm(bases, sb)) == want
    finally:
        eng.msm_configure()
        eng.msm_configure_glv(0)


@pytest.mark.parametrize("kind", ["random", "edges", "equal_scalars"])
@pytest.mark.parametrize("n", [600, 20000])
def test_msm_lean_accumulation(eng, pkg, n, kind):
    """The shipped library's bucket accumulation is k_msm_accumulate_lean (inline-asm Montgomery blocks with fixed temporaries,
    exceptional cases through a fix-up list) and nothing else: the generic kernel (compiler-scheduled C++ formulas, 166 VGPRs)
    is in the measure build only, and the shipped library refuses the debug key that would select it."""
    rng = O.SplitMix64(3050 + n)
    bases, sb, want = _msm_case(rng, n, kind)
    assert norm(eng, eng.g1_msm(bases, sb)) == want
    with pytest.raises(pkg.H2AggError):
        eng.debug_configure("lean_acc", 0)
    eng.debug_configure("lean_acc", 1)


def test_msm_all_zero_scalars_and_identity_bases(eng):
    n = 100
    rng = O.SplitMix64(3100)
    bases = points_from_scalars(rand_frs(rng, n))
    assert norm(eng, eng.g1_msm(bases, bytes(32 * n))) == bytes(64)
    assert norm(eng, eng.g1_msm(bytes(64 * n), fr_bytes(rand_frs(rng, n)))) == bytes(64)


def test_msm_empty_is_the_reference_panic(eng, pkg):
    with pytest.raises(pkg.EmptyMultiExp):
        eng.g1_msm(b"", b"")


def test_msm_noncanonical_scalar_rejected(eng, pkg):
    with pytest.raises(pkg.H2AggError) as ei:
        eng.g1_msm(G_BYTES, (O.R + 5).to_bytes(32, "little"))
    assert ei.value.code == pkg.ERR_NONCANONICAL


def test_msm_preloaded_and_prefix(eng):
    rng = O.SplitMix64(3200)
    n = 500
    ks, ss = rand_frs(rng, n), rand_frs(rng, n)
    bases = points_from_scalars(ks)
    h = eng.bases_upload(bases)
    try:
        assert eng.bases_download(h, 0, n) == bases
        for m in (n, 123):
            want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks[:m], ss[:m])) % O.R, O.G1))
            assert norm(eng, eng.g1_msm_preloaded(h, fr_bytes(ss[:m]))) == want
    finally:
        eng.bases_free(h)


@pytest.mark.parametrize("n", [5, 300, 2000])
def test_eval_flat(eng, n):
    # evaluation.rs:189-200: multi_exp over scalar-carrying entries + add of scalar-less points
    rng = O.SplitMix64(4000 + n)
    pts = points_from_scalars(rand_frs(rng, n))
    ss = fr_bytes(rand_frs(rng, n))
    has = bytes([0 if i % 5 == 1 else 1 for i in range(n)])
    assert norm(eng, eng.eval_flat(pts, ss, has)) == cref.eval_flat(pts, ss, has, n)


def test_eval_flat_without_scalars_is_the_reference_panic(eng, pkg):
    with pytest.raises(pkg.EmptyMultiExp):
        eng.eval_flat(G_BYTES * 2, bytes(64), bytes(2))


@pytest.mark.parametrize("sub_bits,tile", [(0, -1), (0, -2), (4, 0), (12, 0), (9, 256), (0, 4096)])
def test_msm_sort_variants(eng, sub_bits, tile):
    """both sort implementations (LDS-staged, direct) and their knobs give the same group element"""
    rng = O.SplitMix64(5000 + sub_bits)
    n = 5000
    bases, sb, want = _msm_case(rng, n, "edges")
    eng.msm_configure_sort(sub_bits, tile)
    try:
        assert norm(eng, eng.g1_msm(bases, sb)) == want
        eng.msm_configure(window_bits=13)
        assert norm(eng, eng.g1_msm(bases, sb)) == want
    finally:
        eng.msm_configure()
        eng.msm_configure_sort()



def test_instance_commitment(eng, pkg):
    """assign_instance_commitment for one column (verify.rs:601-603, 623-640): sum inst_i * g_lagrange[i] against a
    preloaded table, identity for an empty column, the length bound as an error."""
    rng = O.SplitMix64(7000)
    n_table, blinding = 64, 5
    ks = rand_frs(rng, n_table)
    g_lagrange = points_from_scalars(ks)
    h = eng.bases_upload(g_lagrange)
    try:
        max_len = n_table - (blinding + 1)
        for m in (1, 7, max_len):
            inst = rand_frs(rng, m)
            got = norm(eng, eng.instance_commitment(h, fr_bytes(inst), max_len))
            acc = O.INF
            for i, s in enumerate(inst):                       # the reference loop: scalar_mul_constant + add
                acc = O.add(acc, O.scalar_mul(s, O.aff_from_bytes(g_lagrange[64 * i:64 * i + 64])))
            assert got == O.aff_to_bytes(acc)
        assert norm(eng, eng.instance_commitment(h, b"", max_len)) == bytes(64)
        with pytest.raises(pkg.H2AggError):
            eng.instance_commitment(h, fr_bytes(rand_frs(rng, max_len + 1)), max_len)
    finally:
        eng.bases_free(h)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("glv", [1, -1])
def test_msm_lanes_per_bucket(eng, lanes, glv):
    rng = O.SplitMix64(8000 + lanes)
    n = 4000
    for kind in ("edges", "equal_scalars"):
        bases, sb, want = _msm_case(rng, n, kind)
        eng.msm_configure_lanes_per_bucket(lanes)
        eng.msm_configure_glv(glv)
        eng.msm_configure(big_bucket_threshold=40)
        try:
            assert norm(eng, eng.g1_msm(bases, sb)) == want
        finally:
            eng.msm_configure()
            eng.msm_configure_glv(0)
            eng.msm_configure_lanes_per_bucket(0)


def test_fr_tape_eval(eng, pkg):
    """h2agg_fr_tape_eval: random straight-line Fr programs (mul / add / sub DAGs with long chains and wide levels) against
    exact big-integer evaluation; a lagrange-style expression (x^n - 1) / (n (x - w^i)) numerator chain; error paths."""
    rng = O.SplitMix64(0x7A9E)
    for trial in range(6):
        nconst, nops = 1 + rng.next() % 40, rng.next() % 3000
        vals = [rng.fr() for _ in range(nconst)]
        if trial == 0:
            vals[0] = 0
        regs = list(vals)
        ops = []
        for k in range(nops):
            op = rng.next() % 3
            hi = nconst + k
            a = hi - 1 if (trial % 2 and k) else rng.next() % hi          # odd trials: one long dependency chain
            b = rng.next() % hi
            ops.append((op, a, b))
            regs.append((regs[a] * regs[b], regs[a] + regs[b], regs[a] - regs[b])[op] % O.R)
        outs = [rng.next() % len(regs) for _ in range(1 + rng.next() % 50)] + [len(regs) - 1]
        got = eng.fr_tape_eval(fr_bytes(vals), ops, outs)
        assert got == fr_bytes([regs[i] for i in outs]), trial
    # x^(2^10) - 1 by repeated squaring, then times a constant: the shape of lagrange.rs / vanish.rs
    x = rng.fr()
    ops = [(0, 0, 0)] + [(0, 2 + k, 2 + k) for k in range(9)] + [(2, 11, 1)]
    assert eng.fr_tape_eval(fr_bytes([x, 1]), ops, [12]) == fr_bytes([(pow(x, 1 << 10, O.R) - 1) % O.R])
    assert eng.fr_tape_eval(fr_bytes([5]), [], [0]) == fr_bytes([5])
    with pytest.raises(pkg.H2AggError):
        eng.fr_tape_eval(fr_bytes([1, 2]), [(0, 0, 2)], [2])               # operand not yet defined
    with pytest.raises(pkg.H2AggError):
        eng.fr_tape_eval(fr_bytes([1, 2]), [(7, 0, 1)], [2])               # unknown opcode
    with pytest.raises(pkg.H2AggError):
        eng.fr_tape_eval(fr_bytes([1, 2]), [(0, 0, 1)], [3])               # output out of range
    with pytest.raises(pkg.H2AggError):
        eng.fr_tape_eval(O.R.to_bytes(32, "little"), [], [0])              # non-canonical input


@pytest.mark.parametrize("kind", ["mixed", "wide_levels", "recycled_chain", "too_many_live", "consts_kept_to_the_end"])
def test_fr_tape_register_file_in_lds(eng, pkg, kind):
    """k_tape_run_lds (csrc/schema.hpp): the tape's values live in LDS slots handed out by the host's liveness allocator.
    Programs that stress it — levels wider than the workgroup, a long chain whose dead values' slots are reused thousands of
    times, more live values than slots (must fall back to the L2 register file), constants read for the first time at the very
    end, inversions — against exact big-integer evaluation and against the same program with debug key tape_lds = 0."""
    rng = O.SplitMix64(0x7D5 + len(kind))
    if kind == "mixed":
        nconst, nops = 300, 6000
    elif kind == "wide_levels":
        nconst, nops = 3000, 3500          # every op reads two constants: ONE level of 3500 operations (strip-mined by 1024 lanes)
    elif kind == "recycled_chain":
        nconst, nops = 5, 9000
    elif kind == "too_many_live":
        nconst, nops = 5000, 5001          # all 5000 constants stay live until the last operations
    else:
        nconst, nops = 3900, 3900
    vals = [rng.fr() or 1 for _ in range(nconst)]
    regs, ops = list(vals), []
    for k in range(nops):
        hi = nconst + k
        op = rng.next() % 3
        if kind == "mixed":
            a = hi - 1 if (k % 3 == 0 and k) else rng.next() % hi
            b = rng.next() % hi
            if k % 977 == 5:
                op, b = 3, a                # an inversion (b unused)
        elif kind == "wide_levels":
            a, b = rng.next() % nconst, rng.next() % nconst
        elif kind == "recycled_chain":
            a, b = hi - 1, (hi - 2 if k > 1 else rng.next() % nconst)
            op = 0 if k % 2 else 1
        elif kind == "too_many_live":
            a, b = (hi - 1 if k else 0), k % nconst
        else:
            a, b = (hi - 1 if k else 0), nconst - 1 - (k % nconst)
        ops.append((op, a, b))
        if op == 3:
            regs.append(pow(regs[a], O.R - 2, O.R) if regs[a] else 0)
        else:
            regs.append((regs[a] * regs[b], regs[a] + regs[b], regs[a] - regs[b])[op] % O.R)
    outs = sorted({rng.next() % len(regs) for _ in range(200)} | {len(regs) - 1, 0, nconst - 1, nconst})
    want = fr_bytes([regs[i] for i in outs])
    assert eng.fr_tape_eval(fr_bytes(vals), ops, outs) == want
    eng.debug_configure("tape_lds", 0)
    try:
        assert eng.fr_tape_eval(fr_bytes(vals), ops, outs) == want
    finally:
        eng.debug_configure("tape_lds", 1)


def test_msm_over_projective_points(eng):
    """h2agg_g1_msm_jac: the points as `Vec<C::CurveExt>` (projective, arbitrary z, some identities) normalised on the
    device; same bytes as normalising first and calling h2agg_g1_msm, and as the oracle's naive multi_exp"""
    from oracle import bn254 as O, cref
    rng = O.SplitMix64(0x1AC)
    for n in (1, 7, 300, 5000):
        ks = [rng.fr() for _ in range(n)]
        aff = cref.g1_batch_to_affine(cref.g1_batch_scalar_mul(O.aff_to_bytes(O.G1) * n, b"".join(O.fe_to_bytes(k) for k in ks), n), n)
        jac = bytearray()
        for i in range(n):
            x = int.from_bytes(aff[64 * i:64 * i + 32], "little")
            y = int.from_bytes(aff[64 * i + 32:64 * i + 64], "little")
            if i % 11 == 5:
                jac += (0).to_bytes(32, "little") + (1).to_bytes(32, "little") + (0).to_bytes(32, "little")   # identity
                aff = aff[:64 * i] + bytes(64) + aff[64 * i + 64:]
                continue
            z = 1 if i % 3 == 0 else rng.next() % O.P or 1
            jac += (x * z * z % O.P).to_bytes(32, "little") + (y * z * z * z % O.P).to_bytes(32, "little") + z.to_bytes(32, "little")
        sc = b"".join(O.fe_to_bytes(rng.fr()) for _ in range(n))
        got = eng.g1_batch_to_affine(eng.g1_msm_jac(bytes(jac), sc))
        assert got == eng.g1_batch_to_affine(eng.g1_msm(aff, sc))
        assert got == cref.multi_exp_naive(aff, sc, n)
    import pytest
    with pytest.raises(Exception):
        eng.g1_msm_jac(b"", b"")
