"""CPU tests (-m "not gpu"): pin the oracle.

The reference has no golden vectors; what pins the oracle:
  * moduli literals the reference carries (verifier.sol:40-41, 143-144, 292),
  * committed fixtures produced by exact big-integer arithmetic (tests/golden/make_golden.py),
  * the reference's own algebraic identities (five_native_ecc.rs:60-240),
  * a public BN254 known answer (EIP-196: 2*G).
"""
import json
import os

import pytest

from oracle import bn254 as O, cref
from oracle import schema as S
from tests.util import G_BYTES, fr_bytes, points_from_scalars, rand_frs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_moduli_match_reference_literals():
    # halo2-snark-aggregator-solidity/templates/verifier.sol:40-41 (q_mod), :143-144 (p_mod), :292 (hex r)
    assert O.R == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert O.P == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert O.R == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    for which, m in ((0, O.R), (1, O.P)):
        mod, r1, r2, inv = cref.constants(which)
        assert mod == m and r1 == (1 << 256) % m and r2 == pow(1 << 256, 2, m)
        assert (inv * m) % (1 << 64) == (1 << 64) - 1


def test_curve_constants_and_public_known_answer():
    assert O.is_on_curve(O.G1)
    assert O.scalar_mul(O.R, O.G1) is O.INF and O.scalar_mul(O.R - 1, O.G1) == O.neg(O.G1)
    # EIP-196 ecAdd/ecMul vector: 2*(1,2)
    assert O.double(O.G1) == (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
                              0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)


@pytest.mark.parametrize("which,name", [(0, "fr"), (1, "fq")])
def test_field_fixtures(which, name):
    kats = load("field_kats.json")[name]
    a = b"".join(bytes.fromhex(k["a"]) for k in kats)
    b = b"".join(bytes.fromhex(k["b"]) for k in kats)
    n = len(kats)
    for op, key in ((cref.OP_ADD, "add"), (cref.OP_SUB, "sub"), (cref.OP_MUL, "mul"), (cref.OP_SQR, "sqr")):
        got = cref.field_batch_op(which, op, a, b if op <= 2 else None, n)
        assert got == b"".join(bytes.fromhex(k[key]) for k in kats), key
    nz = [k for k in kats if k["inv"] is not None]
    got = cref.field_batch_op(which, cref.OP_INV, b"".join(bytes.fromhex(k["a"]) for k in nz), None, len(nz))
    assert got == b"".join(bytes.fromhex(k["inv"]) for k in nz)
    with pytest.raises(ZeroDivisionError):
        cref.field_batch_op(which, cref.OP_INV, bytes(32), None, 1)      # mock/arith/field.rs:113 panics


def test_point_fixtures():
    kats = load("point_kats.json")
    a = b"".join(bytes.fromhex(k["a_jac"]) for k in kats["add"])
    b = b"".join(bytes.fromhex(k["b_jac"]) for k in kats["add"])
    n = len(kats["add"])
    assert cref.g1_batch_to_affine(cref.g1_batch_add(a, b, n), n) == b"".join(bytes.fromhex(k["sum_aff"]) for k in kats["add"])
    assert cref.g1_batch_to_affine(cref.g1_batch_add(a, b, n, True), n) == b"".join(
        bytes.fromhex(k["diff_aff"]) for k in kats["add"])
    sm = kats["scalar_mul"]
    got = cref.g1_batch_scalar_mul(b"".join(bytes.fromhex(k["base_aff"]) for k in sm),
                                   b"".join(bytes.fromhex(k["scalar"]) for k in sm), len(sm))
    assert cref.g1_batch_to_affine(got, len(sm)) == b"".join(bytes.fromhex(k["out_aff"]) for k in sm)


def test_msm_fixtures():
    for k in load("msm_kats.json"):
        bases, scalars, want = bytes.fromhex(k["bases_aff"]), bytes.fromhex(k["scalars"]), bytes.fromhex(k["out_aff"])
        assert cref.multi_exp_naive(bases, scalars, k["n"]) == want
        for c in (3, 7):
            assert cref.msm_pippenger(bases, scalars, k["n"], c, 2) == want
            assert cref.msm_pippenger2(bases, scalars, k["n"], max(c, 2), 3, 2) == want   # signed digits / XYZZ buckets
        assert cref.eval_flat(bases, scalars, bytes([1]) * k["n"], k["n"]) == want


def test_multi_exp_empty_panics_like_the_reference():
    with pytest.raises(ValueError):
        O.multi_exp([], [])
    with pytest.raises(ValueError):
        cref.multi_exp_naive(b"", b"", 0)


def test_reference_identities_five_native_ecc():
    rng = O.SplitMix64(42)
    s1, s2, s3, s4 = (rng.fr() for _ in range(4))
    g = O.G1
    # add: s1*G + s2*G == (s1+s2)*G                                       five_native_ecc.rs:60-88
    assert O.add(O.scalar_mul(s1, g), O.scalar_mul(s2, g)) == O.scalar_mul(s1 + s2, g)
    # mul: (s1*G)*s2 == (s1*s2)*G, incl. zero scalar and identity point     :118-150
    assert O.scalar_mul(s2, O.scalar_mul(s1, g)) == O.scalar_mul(s1 * s2, g)
    assert O.scalar_mul(0, O.scalar_mul(s1, g)) is O.INF and O.scalar_mul(s2, O.INF) is O.INF
    # shamir: [s1 G, s2 G].[s3, s4] == (s1 s3 + s2 s4) G, and the 1-point case   :152-182
    pts = points_from_scalars([s1, s2])
    want = O.aff_to_bytes(O.scalar_mul(s1 * s3 + s2 * s4, g))
    assert cref.multi_exp_naive(pts, fr_bytes([s3, s4]), 2) == want
    assert cref.multi_exp_naive(pts[:64], fr_bytes([s3]), 1) == O.aff_to_bytes(O.scalar_mul(s1 * s3, g))
    # double incl. identity                                                :219-240
    assert O.double(O.INF) is O.INF and O.double(O.scalar_mul(s1, g)) == O.scalar_mul(2 * s1, g)


def test_c_oracle_vs_identity_at_scale():
    rng = O.SplitMix64(43)
    n = 2048
    ks, ss = rand_frs(rng, n), rand_frs(rng, n)
    bases = points_from_scalars(ks)
    want = O.aff_to_bytes(O.scalar_mul(sum(k * s for k, s in zip(ks, ss)) % O.R, O.G1))
    assert cref.msm_pippenger(bases, fr_bytes(ss), n, 8, 4) == want
    for c2, thr, jpt in ((2, 1, 1), (8, 4, 4), (13, 2, 1), (16, 3, 2)):
        assert cref.msm_pippenger2(bases, fr_bytes(ss), n, c2, thr, jpt) == want
    assert cref.multi_exp_naive(bases[:64 * 200], fr_bytes(ss[:200]), 200) == O.aff_to_bytes(
        O.scalar_mul(sum(k * s for k, s in zip(ks[:200], ss[:200])) % O.R, O.G1))


def test_field_chip_defaults():
    sc, ctx = S.OracleFieldChip(), S.OracleCtx()
    rng = O.SplitMix64(44)
    b = rng.fr()
    for e in (1, 2, 3, 5, 8, 1023, 0xFFFFFFFF):
        assert sc.pow_constant(ctx, b, e) == pow(b, e, O.R)               # arith/field.rs:83-104
    with pytest.raises(AssertionError):
        sc.pow_constant(ctx, b, 0)
    v = rand_frs(rng, 9)
    assert sc.mul_add_accumulate(ctx, v, b) == sum(x * pow(b, len(v) - 1 - i, O.R) for i, x in enumerate(v)) % O.R
    with pytest.raises(ZeroDivisionError):
        sc.div(ctx, 3, 0)
    assert str(ctx) == "(total points: 0)"                                # mock/arith/field.rs:17-21


# ------------------------------------------------------------------ schema layer
def _tree(t):
    if t[0] == "C":
        return S.commit(S.CommitQuery(t[1], O.aff_from_bytes(bytes.fromhex(t[2])), None))
    if t[0] == "E":
        return S.evalq(S.CommitQuery("", None, O.fe_from_bytes(bytes.fromhex(t[1]))))
    if t[0] == "S":
        return S.scalar(O.fe_from_bytes(bytes.fromhex(t[1])))
    l, r = _tree(t[1]), _tree(t[2])
    return l + r if t[0] == "+" else l * r


def _expand(t, coeff, acc):
    """independent check: expand the tree algebraically into {key: coeff}, scalars under ''.  Plain
    distributive-law evaluation, no has-commitment bookkeeping."""
    if t[0] == "C":
        acc.setdefault(t[1], [bytes.fromhex(t[2]), 0])
        acc[t[1]][1] = (acc[t[1]][1] + coeff) % O.R
    elif t[0] in ("E", "S"):
        acc.setdefault("", [None, 0])
        acc[""][1] = (acc[""][1] + coeff * O.fe_from_bytes(bytes.fromhex(t[1]))) % O.R
    elif t[0] == "+":
        _expand(t[1], coeff, acc)
        _expand(t[2], coeff, acc)
    else:
        lc, rc = _has_c(t[1]), _has_c(t[2])
        assert not (lc and rc)
        s_side, rem = (t[1], t[2]) if not lc else (t[2], t[1])
        tmp = {}
        _expand(s_side, 1, tmp)
        _expand(rem, coeff * tmp[""][1] % O.R, acc)


def _has_c(t):
    return t[0] == "C" or (t[0] in "+*" and (_has_c(t[1]) or _has_c(t[2])))


def test_schema_fixtures_and_independent_expansion():
    for k in load("schema_kats.json"):
        ctx, sc, pc = S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip()
        proof = S.MultiOpenProof(_tree(k["w_x"]), _tree(k["w_g"]))
        assert str(proof) == k["estimate"]
        left, right, names = S.evaluate_multiopen_proof(ctx, sc, pc, proof)
        assert names == k["names"] and len(ctx.point_list) == k["point_list_len"]
        assert S.final_pair_bytes(left, right).hex() == k["final_pair"]
        # independent algebra: W = sum coeff_k P_k (+/-) e*G
        for tree, sign, got in ((k["w_x"], 1, left), (k["w_g"], -1, right)):
            acc = {}
            _expand(tree, 1, acc)
            e = acc.pop("", [None, 0])[1]
            keys = list(acc)
            pts = b"".join(acc[q][0] for q in keys)
            want = O.aff_from_bytes(cref.multi_exp_naive(pts, fr_bytes([acc[q][1] for q in keys]), len(keys)))
            want = O.add(want, O.scalar_mul(sign * e, O.G1))
            assert got == want


def test_schema_eval_semantics_small():
    ctx, sc, pc = S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip()
    P, Q = O.scalar_mul(5, O.G1), O.scalar_mul(9, O.G1)
    cp, cq = S.CommitQuery("p", P, 11), S.CommitQuery("q", Q, 13)
    # (C_p + e_p) * 3 + (C_q + e_q) + C_p : key merge, scalar-less point, eval accumulation
    s = (S.commit(cp) + S.evalq(cp)) * S.scalar(3) + (S.commit(cq) + S.evalq(cq)) + S.commit(cp)
    pt, e, names = s.eval(ctx, sc, pc, 1)
    assert names == ["p", "", "q"]
    assert e == (3 * 11 + 13) % O.R
    assert pt == O.add(O.scalar_mul(4, P), Q)          # p: 3 + implicit 1 (None == one, evaluation.rs:251-261)
    assert s.estimate() == 4 and str(ctx) == "(total points: 1)"
    # a schema whose only commitment has no scalar hits the reference panic (multi_exp of zero pairs)
    with pytest.raises(ValueError):
        S.commit(cp).eval(ctx, sc, pc, 1)


def test_point_wire_format_roundtrip():
    """compress / decompress (the proof wire format, transcript.rs:56-79) are inverse on curve points, reject
    non-residues and non-canonical x, and map 32 zero bytes to the identity."""
    rng = O.SplitMix64(0x77)
    for _ in range(40):
        p = O.scalar_mul(rng.fr(), O.G1)
        enc = O.compress(p)
        assert len(enc) == 32 and O.decompress(enc) == p and O.is_on_curve(p)
        assert O.decompress(O.compress(O.neg(p))) == O.neg(p) and O.compress(O.neg(p))[:31] == enc[:31]
    assert O.compress(O.INF) == bytes(32) and O.decompress(bytes(32)) is O.INF
    assert O.decompress(O.compress(O.G1)) == (1, 2) and O.compress(O.G1)[31] == 0      # y = 2 is even
    bad = 0
    for x in range(2, 60):
        try:
            O.decompress(x.to_bytes(32, "little"))
        except ValueError:
            bad += 1
    assert 15 < bad < 45                                                                 # about half are non-residues
    with pytest.raises(ValueError):
        O.decompress((O.P + 1).to_bytes(32, "little"))
    with pytest.raises(ValueError):
        O.decompress(bytes(31) + bytes([0x80]))                                          # x = 0 with the sign bit: 3 is a non-residue
