"""GPU: the trait-level mirror (chips.GpuFieldChip / GpuEccChip / GpuChipCtx over libh2agg.so) against the oracle's Mock
chips — every method of ArithCommonChip / ArithFieldChip / ArithEccChip (arith/{common,field,ecc}.rs), and the oracle's
GENERIC eval / evaluate_multiopen_proof code run once with each pair of chips (the reference's own structure: the same
verifier code over different chip implementations, api/src/tests/systems/halo2/add_mul_test/verify_aggregation.rs)."""
import importlib

import pytest

import __graft_entry__ as entry
from oracle import bn254 as O
from oracle import schema as S
from tests.golden.make_golden import synthetic_proof

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chips(eng, pkg):
    return importlib.import_module(entry.PKG_NAME + ".chips")


def fe(x):
    return O.fe_to_bytes(x % O.R)


def jac_of(chips, pc, p):
    return pc.assign_const(None, O.aff_to_bytes(p))


def test_field_chip_methods(eng, chips):
    g, o = chips.GpuFieldChip(eng), S.OracleFieldChip()
    ctx, octx = chips.GpuChipCtx(), S.OracleCtx()
    rng = O.SplitMix64(0xC41)
    a, b, c = rng.fr(), rng.fr() or 1, rng.fr()
    assert g.add(ctx, fe(a), fe(b)) == fe(o.add(octx, a, b))
    assert g.sub(ctx, fe(a), fe(b)) == fe(o.sub(octx, a, b))
    assert g.mul(ctx, fe(a), fe(b)) == fe(o.mul(octx, a, b))
    assert g.div(ctx, fe(a), fe(b)) == fe(o.div(octx, a, b))
    assert g.square(ctx, fe(a)) == fe(o.square(octx, a))
    assert g.mul_add_constant(ctx, fe(a), fe(b), fe(c)) == fe(o.mul_add_constant(octx, a, b, c))
    assert g.mul_add(ctx, fe(a), fe(b), fe(c)) == fe(o.mul_add(octx, a, b, c))
    xs, cs = [rng.fr() for _ in range(9)], [rng.fr() for _ in range(9)]
    assert g.sum_with_coeff_and_constant(ctx, [(fe(x), fe(k)) for x, k in zip(xs, cs)], fe(c)) == \
        fe(o.sum_with_coeff_and_constant(octx, list(zip(xs, cs)), c))
    assert g.sum_with_coeff_and_constant(ctx, [], fe(c)) == fe(c)
    assert g.sum_with_constant(ctx, [fe(x) for x in xs], fe(c)) == fe(o.sum_with_constant(octx, xs, c))
    assert g.mul_add_accumulate(ctx, [fe(x) for x in xs], fe(b)) == fe(o.mul_add_accumulate(octx, xs, b))
    assert g.mul_add_accumulate(ctx, [], fe(b)) == fe(0)
    for e in (1, 2, 3, 5, 8, 255, 1 << 17, (1 << 20) - 3):
        assert g.pow_constant(ctx, fe(a), e) == fe(o.pow_constant(octx, a, e))
    assert g.assign_zero(ctx) == fe(0) and g.assign_one(ctx) == fe(1)
    assert g.to_value(g.normalize(ctx, g.assign_const(ctx, fe(a)))) == fe(a)
    with pytest.raises(ZeroDivisionError):                  # b.invert().unwrap()  mock/arith/field.rs:113
        g.div(ctx, fe(a), fe(0))


def test_ecc_chip_methods(eng, chips, pkg):
    pc, oc = chips.GpuEccChip(eng), S.OracleEccChip()
    ctx, octx = chips.GpuChipCtx(), S.OracleCtx()
    rng = O.SplitMix64(0xC42)
    P, Q = O.scalar_mul(rng.fr(), O.G1), O.scalar_mul(rng.fr(), O.G1)
    jp, jq = jac_of(chips, pc, P), jac_of(chips, pc, Q)
    aff = lambda j: pc.to_value(j)                                                       # noqa: E731
    assert aff(pc.add(ctx, jp, jq)) == O.aff_to_bytes(oc.add(octx, P, Q))
    assert aff(pc.sub(ctx, jp, jq)) == O.aff_to_bytes(oc.sub(octx, P, Q))
    assert aff(pc.add(ctx, jp, jp)) == O.aff_to_bytes(O.add(P, P))                       # P + P
    assert aff(pc.sub(ctx, jp, jp)) == bytes(64)                                         # P - P
    assert aff(pc.add(ctx, jp, pc.assign_zero(ctx))) == O.aff_to_bytes(P)
    assert aff(pc.assign_one(ctx)) == O.aff_to_bytes(O.G1)
    assert aff(pc.assign_const(ctx, bytes(64))) == bytes(64)
    s = rng.fr()
    assert aff(pc.scalar_mul(ctx, fe(s), jp)) == O.aff_to_bytes(oc.scalar_mul(octx, s, P))
    assert aff(pc.scalar_mul_constant(ctx, fe(s), O.aff_to_bytes(Q))) == O.aff_to_bytes(oc.scalar_mul_constant(octx, s, Q))
    assert aff(pc.scalar_mul(ctx, fe(0), jp)) == bytes(64)
    pts = [O.scalar_mul(rng.fr(), O.G1) for _ in range(12)] + [O.INF]
    scs = [rng.fr() for _ in range(12)] + [rng.fr()]
    got = pc.multi_exp(ctx, [jac_of(chips, pc, p) for p in pts], [fe(x) for x in scs])
    assert aff(got) == O.aff_to_bytes(oc.multi_exp(octx, pts, scs))
    assert str(ctx) == str(octx) == "(total points: 13)" and ctx.point_list == octx.point_list
    with pytest.raises(pkg.EmptyMultiExp):                  # acc.unwrap()  mock/arith/ecc.rs:128
        pc.multi_exp(ctx, [], [])


def _to_gpu_values(chips, pc, s):
    """an oracle Schema with Python ints / point tuples -> the same Schema over the GPU chips' value types"""
    if s.kind == "commitment":
        return S.commit(S.CommitQuery(s.cq.key, jac_of(chips, pc, s.cq.commitment), None))
    if s.kind == "eval":
        return S.evalq(S.CommitQuery("", None, fe(s.cq.eval)))
    if s.kind == "scalar":
        return S.scalar(fe(s.s))
    l, r = _to_gpu_values(chips, pc, s.l), _to_gpu_values(chips, pc, s.r)
    return l + r if s.kind == "add" else l * r


def test_generic_verifier_code_over_gpu_chips(eng, chips):
    """the oracle's chip-generic evaluate_multiopen_proof (restating verify.rs:705-731 + evaluation.rs:172-293), run once
    over the Mock-style oracle chips and once over the GPU chips: same final pair, names and ctx.point_list length"""
    rng = O.SplitMix64(0xC43)
    proofs = []
    for i in range(3):
        sp = synthetic_proof(rng, "circuit_p%d" % i, 2, 2, 3, 2)
        proofs.append(S.batch_multi_open_proofs(sp["key"], sp["queries"], sp["w"], sp["v"], sp["u"]))
    agg = S.aggregate_fold(proofs, rng.fr())
    octx = S.OracleCtx()
    want_l, want_r, want_names = S.evaluate_multiopen_proof(octx, S.OracleFieldChip(), S.OracleEccChip(), agg)
    sc, pc, ctx = chips.GpuFieldChip(eng), chips.GpuEccChip(eng), chips.GpuChipCtx()
    gagg = S.MultiOpenProof(_to_gpu_values(chips, pc, agg.w_x), _to_gpu_values(chips, pc, agg.w_g))
    left, right, names = S.evaluate_multiopen_proof(ctx, sc, pc, gagg)
    assert left + right == S.final_pair_bytes(want_l, want_r)
    assert names == want_names and len(ctx.point_list) == len(octx.point_list)
