"""GPU parity of the EvaluationQuerySchema entry point (evaluation.rs:172-293, verify.rs:705-731) against the
oracle restatement and the committed schema fixtures.  Reads like the reference's Mock-chip runs
(halo2-snark-aggregator-api/src/tests/systems/halo2/add_mul_test/verify_aggregation.rs:32-149): build the
aggregated MultiOpenProof schema, evaluate it, compare the final pair."""
import json
import os

import pytest

from oracle import bn254 as O
from oracle import schema as S
from tests.golden.make_golden import synthetic_proof

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def build(pkg, b, t):
    """fixture tree -> GPU-backend schema"""
    if t[0] == "C":
        return b.commit(pkg.CommitQuery(t[1], bytes.fromhex(t[2]), None))
    if t[0] == "E":
        return b.evalq(pkg.CommitQuery("", None, bytes.fromhex(t[1])))
    if t[0] == "S":
        return b.scalar(bytes.fromhex(t[1]))
    l, r = build(pkg, b, t[1]), build(pkg, b, t[2])
    return l + r if t[0] == "+" else l * r


def mirror(pkg, b, s):
    """oracle schema tree -> GPU-backend schema (same shape)"""
    if s.kind == "commitment":
        return b.commit(pkg.CommitQuery(s.cq.key, O.aff_to_bytes(s.cq.commitment), None))
    if s.kind == "eval":
        return b.evalq(pkg.CommitQuery("", None, O.fe_to_bytes(s.cq.eval)))
    if s.kind == "scalar":
        return b.scalar(O.fe_to_bytes(s.s))
    l, r = mirror(pkg, b, s.l), mirror(pkg, b, s.r)
    return l + r if s.kind == "add" else l * r


def test_schema_fixtures(eng, pkg):
    for k in load("schema_kats.json"):
        b = pkg.SchemaBuilder(eng)
        w_x, w_g = build(pkg, b, k["w_x"]), build(pkg, b, k["w_g"])
        assert "(estimated scalar mult of points: %d)" % (w_x.estimate() + w_g.estimate()) == k["estimate"]
        left, right, names = b.evaluate_multiopen_proof(w_x, w_g)
        assert (left + right).hex() == k["final_pair"]            # fs.rs:187-190 layout
        assert names == k["names"]
        assert b.point_list_len() == k["point_list_len"]          # MockChipCtx Display (mock/arith/field.rs:17-21)
        b.close()


def test_schema_eval_semantics_small(eng, pkg):
    b = pkg.SchemaBuilder(eng)
    P, Q = O.scalar_mul(5, O.G1), O.scalar_mul(9, O.G1)
    cp = pkg.CommitQuery("p", O.aff_to_bytes(P), O.fe_to_bytes(11))
    cq = pkg.CommitQuery("q", O.aff_to_bytes(Q), O.fe_to_bytes(13))
    s = (b.commit(cp) + b.evalq(cp)) * b.scalar(O.fe_to_bytes(3)) + (b.commit(cq) + b.evalq(cq)) + b.commit(cp)
    jac, e, names = s.eval()
    assert names == ["p", "", "q"]
    assert e == O.fe_to_bytes((3 * 11 + 13) % O.R)
    assert eng.g1_batch_to_affine(jac) == O.aff_to_bytes(O.add(O.scalar_mul(4, P), Q))
    assert s.estimate() == 4 and b.point_list_len() == 1
    # scalar-only schema: no commitment -> multi_exp of zero pairs -> the reference panics
    with pytest.raises(pkg.EmptyMultiExp):
        b.commit(cp).eval()
    # Mul with commitments on both sides: assert_eq!(s.len(), 1) (evaluation.rs:282)
    with pytest.raises(pkg.H2AggError):
        (b.commit(cp) * b.commit(cq)).eval()
    b.close()


@pytest.mark.parametrize("nproofs,shape", [(4, (2, 2, 4, 2)), (8, (6, 4, 5, 3))])
def test_aggregation_vs_oracle(eng, pkg, nproofs, shape):
    """N synthetic proofs folded with lambda (verify.rs:926-938), evaluated on the GPU vs the oracle chips."""
    rng = O.SplitMix64(0xA66 + nproofs)
    proofs = []
    for i in range(nproofs):
        sp = synthetic_proof(rng, "circuit_p%d" % i, *shape)
        proofs.append(S.batch_multi_open_proofs(sp["key"], sp["queries"], sp["w"], sp["v"], sp["u"]))
    agg = S.aggregate_fold(proofs, rng.fr())
    ctx, sc, pc = S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip()
    want_l, want_r, want_names = S.evaluate_multiopen_proof(ctx, sc, pc, agg)
    b = pkg.SchemaBuilder(eng)
    left, right, names = b.evaluate_multiopen_proof(mirror(pkg, b, agg.w_x), mirror(pkg, b, agg.w_g))
    assert left + right == S.final_pair_bytes(want_l, want_r)
    assert names == want_names and b.point_list_len() == len(ctx.point_list)
    b.close()


def test_cpp_batch_builders_match_oracle(eng, pkg):
    """h2agg_schema_evaluation_queries + h2agg_schema_batch_multi_open (multiopen.rs:23-102 in the C++ host
    layer) against the oracle's batch_multi_open_proofs on the same simple queries; also the w-count assert."""
    rng = O.SplitMix64(0xB17C)
    nproofs = 3
    want_proofs, specs = [], []
    for i in range(nproofs):
        key = "c_p%d" % i
        x = rng.fr()
        rp = {0: x, 1: x * 7 % O.R, -6: x * 11 % O.R}
        rots = [0] * 5 + [1, 0, -6, 1, 0, -6, 0]
        spec = [(rot, "%s_q%d" % (key, k), rp[rot], O.scalar_mul(rng.fr(), O.G1), rng.fr()) for k, rot in enumerate(rots)]
        w = [O.scalar_mul(rng.fr(), O.G1) for _ in range(3)]
        v, u = rng.fr(), rng.fr()
        qs = [S.evaluation_query(rot, k, z, c, e) for rot, k, z, c, e in spec]
        want_proofs.append(S.batch_multi_open_proofs(key, qs, w, v, u))
        specs.append((key, spec, w, v, u))
    lam = rng.fr()
    agg = S.aggregate_fold(want_proofs, lam)
    want_l, want_r, want_names = S.evaluate_multiopen_proof(S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip(), agg)
    b = pkg.SchemaBuilder(eng)
    got = []
    for key, spec, w, v, u in specs:
        qn = b.evaluation_queries([k for _r, k, _z, _c, _e in spec], b"".join(O.aff_to_bytes(c) for *_x, c, _e in spec),
                                  b"".join(O.fe_to_bytes(e) for *_x, e in spec))
        w_x, w_g = b.batch_multi_open(key, [r for r, *_x in spec], b"".join(O.fe_to_bytes(z) for _r, _k, z, _c, _e in spec),
                                      qn, b"".join(O.aff_to_bytes(p) for p in w), O.fe_to_bytes(v), O.fe_to_bytes(u))
        got.append((w_x, w_g))
    lam_b = O.fe_to_bytes(lam)
    acc_x, acc_g = got[0]
    for w_x, w_g in got[1:]:
        acc_x, acc_g = acc_x * b.scalar(lam_b) + w_x, acc_g * b.scalar(lam_b) + w_g     # verify.rs:926-938
    assert acc_x.estimate() + acc_g.estimate() == agg.w_x.estimate() + agg.w_g.estimate()
    left, right, names = b.evaluate_multiopen_proof(acc_x, acc_g)
    assert left + right == S.final_pair_bytes(want_l, want_r) and names == want_names
    # wrong number of W commitments: the reference's assert_eq! (multiopen.rs:48)
    key, spec, w, v, u = specs[0]
    qn = b.evaluation_queries([k for _r, k, _z, _c, _e in spec], b"".join(O.aff_to_bytes(c) for *_x, c, _e in spec),
                              b"".join(O.fe_to_bytes(e) for *_x, e in spec))
    with pytest.raises(pkg.H2AggError):
        b.batch_multi_open(key, [r for r, *_x in spec], b"".join(O.fe_to_bytes(z) for _r, _k, z, _c, _e in spec),
                           qn, b"".join(O.aff_to_bytes(p) for p in w[:2]), O.fe_to_bytes(v), O.fe_to_bytes(u))
    b.close()


def _rand_scalar_tree(rng, depth, consts):
    """commitment-free subtree: always prepares to exactly one entry"""
    r = rng.next() % 8
    if depth <= 0 or r < 3:
        v = consts[rng.next() % len(consts)]
        if rng.next() & 1:
            return S.scalar(v)
        return S.evalq(S.CommitQuery("", None, v))
    l, rr = _rand_scalar_tree(rng, depth - 1, consts), _rand_scalar_tree(rng, depth - 1, consts)
    return l + rr if r < 6 else l * rr


def _rand_tree(rng, depth, keys, consts):
    """subtree with at least one commitment; Mul never has commitments on both sides (evaluation.rs:282)"""
    r = rng.next() % 10
    if depth <= 0 or r < 2:
        k, p = keys[rng.next() % len(keys)]
        return S.commit(S.CommitQuery(k, p, None))
    if r < 5:
        return _rand_tree(rng, depth - 1, keys, consts) + _rand_tree(rng, depth - 1, keys, consts)
    if r < 6:
        return _rand_tree(rng, depth - 1, keys, consts) + _rand_scalar_tree(rng, 2, consts)
    if r < 7:
        return _rand_scalar_tree(rng, 2, consts) + _rand_tree(rng, depth - 1, keys, consts)
    if r < 9:
        return _rand_scalar_tree(rng, 2, consts) * _rand_tree(rng, depth - 1, keys, consts)
    return _rand_tree(rng, depth - 1, keys, consts) * _rand_scalar_tree(rng, 2, consts)


@pytest.mark.parametrize("seed", range(12))
def test_random_schema_trees_vs_oracle(eng, pkg, seed):
    """Differential test of the tape recorder (interned constants, power chains, deferred sums: csrc/schema.hpp)
    on random ASTs: repeated keys (merging), repeated constants, scalar-only subtrees on either side of Add / Mul,
    and a long Horner chain in one repeated scalar on top."""
    rng = O.SplitMix64(0x5EED00 + seed)
    keys = [("k%d" % i, O.scalar_mul(rng.fr(), O.G1)) for i in range(1 + seed % 7)]
    consts = [rng.fr() for _ in range(1 + seed % 4)] + [1, 0, O.R - 1][:seed % 4]
    tree = _rand_tree(rng, 2 + seed % 5, keys, consts)
    v = consts[0]
    for j in range(5 * (seed % 6)):                   # v*(v*(...) + q) + q: multiopen.rs:56-60 shape
        k, p = keys[rng.next() % len(keys)]
        q = S.commit(S.CommitQuery(k, p, None)) + S.evalq(S.CommitQuery("", None, rng.fr()))
        tree = S.scalar(v) * tree + q
    ctx, sc, pc = S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip()
    b = pkg.SchemaBuilder(eng)
    g = mirror(pkg, b, tree)
    assert g.estimate() == tree.estimate()
    try:
        want_p, want_e, want_names = tree.eval(ctx, sc, pc, sc.assign_one(ctx))
    except Exception:
        with pytest.raises(pkg.H2AggError):
            g.eval()
        b.close()
        return
    jac, e, names = g.eval()
    assert names == want_names
    assert e == (None if want_e is None else O.fe_to_bytes(want_e))
    assert eng.g1_batch_to_affine(jac) == O.aff_to_bytes(want_p)
    assert b.point_list_len() == len(ctx.point_list)
    b.close()


def test_evaluate_multiopen_prepare(eng, pkg):
    """h2agg_evaluate_multiopen_prepare: the host half ahead of time.  Same pair and names as the plain call; a
    commitment replaced BETWEEN prepare and the evaluation (the instance commitment the device was still computing) is
    the one that counts; a schema changed after the prepare is evaluated from scratch; errors surface at prepare."""
    rng = O.SplitMix64(0x9E9A)

    def build(b, first_commit):
        """two proofs folded with lambda; returns (acc_x, acc_g, first query node, oracle pair, oracle names)"""
        got, want_proofs = [], []
        q0 = None
        r = O.SplitMix64(0x51DE)
        for i in range(2):
            key = "p%d" % i
            x = r.fr()
            rp = {0: x, 1: x * 5 % O.R}
            rots = [0, 0, 1, 0, 1]
            spec = [(rot, "%s_q%d" % (key, k), rp[rot], O.scalar_mul(r.fr(), O.G1), r.fr()) for k, rot in enumerate(rots)]
            if i == 0:
                spec[0] = spec[0][:3] + (first_commit,) + spec[0][4:]
            w = [O.scalar_mul(r.fr(), O.G1) for _ in range(2)]
            v, u = r.fr(), r.fr()
            want_proofs.append(S.batch_multi_open_proofs(key, [S.evaluation_query(*q) for q in spec], w, v, u))
            qn = b.evaluation_queries([k for _r, k, _z, _c, _e in spec], b"".join(O.aff_to_bytes(c) for *_x, c, _e in spec),
                                      b"".join(O.fe_to_bytes(e) for *_x, e in spec))
            if i == 0:
                q0 = qn[0]
            got.append(b.batch_multi_open(key, [rr for rr, *_x in spec], b"".join(O.fe_to_bytes(z) for _r, _k, z, _c, _e in spec),
                                          qn, b"".join(O.aff_to_bytes(p) for p in w), O.fe_to_bytes(v), O.fe_to_bytes(u)))
        lam = r.fr()
        agg = S.aggregate_fold(want_proofs, lam)
        want = S.evaluate_multiopen_proof(S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip(), agg)
        lam_b = O.fe_to_bytes(lam)
        acc_x, acc_g = got[0]
        acc_x, acc_g = acc_x * b.scalar(lam_b) + got[1][0], acc_g * b.scalar(lam_b) + got[1][1]
        return acc_x, acc_g, q0, S.final_pair_bytes(want[0], want[1]), want[2]

    real, placeholder = O.scalar_mul(rng.fr(), O.G1), O.scalar_mul(rng.fr(), O.G1)
    # 1. prepare, then evaluate: equal to the plain call
    b1 = pkg.SchemaBuilder(eng)
    ax, ag, _q0, want_pair, want_names = build(b1, real)
    b1.evaluate_multiopen_prepare(ax, ag)
    left, right, names = b1.evaluate_multiopen_proof(ax, ag)
    assert left + right == want_pair and names == want_names
    # ... and once more on the same schema (the first evaluation left the tape as the preparation found it)
    left, right, names = b1.evaluate_multiopen_proof(ax, ag)
    assert left + right == want_pair and names == want_names
    # 2. built around a placeholder, prepared, THEN the real commitment patched in
    b2 = pkg.SchemaBuilder(eng)
    ax, ag, q0, _wrong_pair, _n = build(b2, placeholder)
    b2.evaluate_multiopen_prepare(ax, ag)
    b2.query_set_commitment(q0, O.aff_to_bytes(real))
    left, right, names = b2.evaluate_multiopen_proof(ax, ag)
    assert left + right == want_pair and names == want_names
    # 3. the schema grows after the prepare: the evaluation of the NEW roots does its own host half
    b3 = pkg.SchemaBuilder(eng)
    ax, ag, _q0, _p, _n = build(b3, real)
    b3.evaluate_multiopen_prepare(ax, ag)
    two = b3.scalar(O.fe_to_bytes(2))
    l2, r2, _ = b3.evaluate_multiopen_proof(ax * two, ag * two)
    dbl = lambda aff: O.aff_to_bytes(O.add(O.aff_from_bytes(aff), O.aff_from_bytes(aff)))   # noqa: E731
    assert l2 == dbl(want_pair[:64]) and r2 == dbl(want_pair[64:])
    # 4. the host half's errors come out of prepare: a Mul of two commitment-carrying sides (evaluation.rs:282)
    b4 = pkg.SchemaBuilder(eng)
    cp = b4.commit(pkg.CommitQuery("a", O.aff_to_bytes(real)))
    with pytest.raises(pkg.H2AggError):
        b4.evaluate_multiopen_prepare(cp * cp, cp)
