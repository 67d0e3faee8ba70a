#!/usr/bin/env python3
"""Generate tests/golden/*.json from the pure-Python big-integer oracle (oracle/bn254.py, oracle/schema.py).

The reference holds no golden vectors for this path and cannot be run here (Rust, unvendored deps), so
these fixtures are the committed known answers: produced by exact big-integer arithmetic written from
the mathematics, then used to pin the C restatement (oracle/bn254_ref.c) and the HIP kernels.

    python tests/golden/make_golden.py        (re-creates the files deterministically)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bn254 as O  # noqa: E402
from oracle import schema as S  # noqa: E402


def hx(b: bytes) -> str:
    return b.hex()


def field_kats():
    rng = O.SplitMix64(0xF1E1D)
    out = {"fr": [], "fq": []}
    for name, m in (("fr", O.R), ("fq", O.P)):
        vals = [(0, 1), (1, 1), (m - 1, m - 1), (m - 1, 2), (2, (m + 1) // 2)]
        for _ in range(11):
            vals.append((int.from_bytes(rng.bytes(64), "little") % m, int.from_bytes(rng.bytes(64), "little") % m))
        for a, b in vals:
            out[name].append({
                "a": hx(O.fe_to_bytes(a)), "b": hx(O.fe_to_bytes(b)),
                "add": hx(O.fe_to_bytes((a + b) % m)), "sub": hx(O.fe_to_bytes((a - b) % m)),
                "mul": hx(O.fe_to_bytes(a * b % m)), "sqr": hx(O.fe_to_bytes(a * a % m)),
                "inv": hx(O.fe_to_bytes(O.inv(a, m))) if a % m else None,
            })
    return out


def point_kats():
    rng = O.SplitMix64(0x9017)
    ks = [rng.fr() for _ in range(6)]
    pts = [O.scalar_mul(k, O.G1) for k in ks]
    P, Q = pts[0], pts[1]
    cases = [("P+Q", P, Q), ("P+P", P, P), ("P+(-P)", P, O.neg(P)), ("inf+P", O.INF, P), ("P+inf", P, O.INF),
             ("inf+inf", O.INF, O.INF), ("G+G", O.G1, O.G1), ("Q+P", Q, P)]
    out = {"add": [], "scalar_mul": []}
    for name, a, b in cases:
        za, zb = rng.fr() % O.P or 1, rng.fr() % O.P or 1
        out["add"].append({"name": name, "a_jac": hx(O.jac_to_bytes(a, za)), "b_jac": hx(O.jac_to_bytes(b, zb)),
                           "sum_aff": hx(O.aff_to_bytes(O.add(a, b))), "diff_aff": hx(O.aff_to_bytes(O.sub(a, b)))})
    for s, base in [(0, P), (1, P), (2, O.G1), (O.R - 1, Q), (rng.fr(), pts[2]), (rng.fr(), O.INF), (5, O.G1)]:
        out["scalar_mul"].append({"base_aff": hx(O.aff_to_bytes(base)), "scalar": hx(O.fe_to_bytes(s)),
                                  "out_aff": hx(O.aff_to_bytes(O.scalar_mul(s, base)))})
    return out


def msm_kats():
    out = []
    for n in (1, 2, 3, 17, 64):
        rng = O.SplitMix64(0x3530 + n)
        ks = [rng.fr() for _ in range(n)]
        ss = [rng.fr() for _ in range(n)]
        if n >= 17:
            ss[0], ss[1], ss[2] = 0, O.R - 1, 1
            ks[3] = ks[4]
            ss[3] = ss[4]                  # duplicate base, same scalar: P + P inside a bucket
            ks[5] = (-ks[6]) % O.R
            ss[5] = ss[6]                  # P and -P with the same scalar
        pts = [O.scalar_mul(k, O.G1) for k in ks]
        if n >= 17:
            pts[7] = O.INF                 # identity base
        res = O.multi_exp(pts, ss)
        tot = sum(k * s for i, (k, s) in enumerate(zip(ks, ss)) if not (n >= 17 and i == 7)) % O.R
        assert res == O.scalar_mul(tot, O.G1)
        out.append({"n": n, "bases_aff": hx(b"".join(O.aff_to_bytes(p) for p in pts)),
                    "scalars": hx(b"".join(O.fe_to_bytes(s) for s in ss)), "out_aff": hx(O.aff_to_bytes(res))})
    return out


def synthetic_proof(rng, key, n_adv, n_fixed, n_perm, n_h):
    """A shape-faithful synthetic multi-open instance: queries at
    rotations {0, +1, -6}, commitments with known discrete logs, random evals and challenges."""
    def pt():
        return O.scalar_mul(rng.fr(), O.G1)
    x = rng.fr()
    omega = rng.fr()
    rot_point = {0: x, 1: x * omega % O.R, -6: x * O.inv(pow(omega, 6, O.R), O.R) % O.R}
    queries = []
    queries.append((0, "%s_instance_commitments0" % key, pt(), rng.fr()))
    for i in range(n_adv):
        queries.append((0, "%s_advice_commitments%d" % (key, i), pt(), rng.fr()))
    perm_pts = [pt() for _ in range(n_perm)]
    for i in range(n_perm):
        k = "%s_0_permutation_product_commitment_%d" % (key, i)
        queries.append((0, k, perm_pts[i], rng.fr()))
        queries.append((1, k, perm_pts[i], rng.fr()))
    for i in reversed(range(n_perm - 1)):
        queries.append((-6, "%s_0_permutation_product_commitment_%d" % (key, i), perm_pts[i], rng.fr()))
    for i in range(n_fixed):
        queries.append((0, "%s_fixed_commitments%d" % (key, i), pt(), rng.fr()))
    for i in range(n_perm):
        queries.append((0, "%s_permutation_commitments%d" % (key, i), pt(), rng.fr()))
    qs = [S.evaluation_query(rot, k, rot_point[rot], c, e) for rot, k, c, e in queries]
    # vanishing: sum_i xn^i [h_i] + expected_h_eval (vanish.rs:56-72)
    xn = rng.fr()
    h = None
    for i in range(n_h):
        cq = S.CommitQuery("%s_h_commitment%d" % (key, i), pt(), None)
        h = S.commit(cq) if h is None else S.scalar(xn) * h + S.commit(cq)
    qs.append((0, x, h + S.scalar(rng.fr())))
    qs.append(S.evaluation_query(0, "%s_random_commitment" % key, x, pt(), rng.fr()))
    w = [pt() for _ in range(3)]
    return dict(key=key, queries=qs, w=w, v=rng.fr(), u=rng.fr())


def flatten_for_fixture(schema):
    """serialise a schema tree to nested lists (JSON)"""
    k = schema.kind
    if k == "commitment":
        return ["C", schema.cq.key, hx(O.aff_to_bytes(schema.cq.commitment))]
    if k == "eval":
        return ["E", hx(O.fe_to_bytes(schema.cq.eval))]
    if k == "scalar":
        return ["S", hx(O.fe_to_bytes(schema.s))]
    return ["+" if k == "add" else "*", flatten_for_fixture(schema.l), flatten_for_fixture(schema.r)]


def schema_kats():
    out = []
    for nproofs, shape in ((1, (2, 2, 4, 2)), (2, (2, 2, 4, 2)), (3, (5, 3, 3, 3))):
        rng = O.SplitMix64(0x5C4E + nproofs)
        proofs = []
        for i in range(nproofs):
            sp = synthetic_proof(rng, "circuit_p%d" % i, *shape)
            proofs.append(S.batch_multi_open_proofs(sp["key"], sp["queries"], sp["w"], sp["v"], sp["u"]))
        lam = rng.fr()
        agg = S.aggregate_fold(proofs, lam)
        ctx, sc, pc = S.OracleCtx(), S.OracleFieldChip(), S.OracleEccChip()
        wx_tree, wg_tree = flatten_for_fixture(agg.w_x), flatten_for_fixture(agg.w_g)
        est = str(agg)
        left, right, names = S.evaluate_multiopen_proof(ctx, sc, pc, agg)
        out.append({"nproofs": nproofs, "w_x": wx_tree, "w_g": wg_tree, "estimate": est,
                    "point_list_len": len(ctx.point_list), "names": names,
                    "final_pair": hx(S.final_pair_bytes(left, right))})
    return out


def wire_kats():
    """proof wire format of G1 points (oracle.bn254.compress / decompress; transcript.rs:56-79)"""
    rng = O.SplitMix64(0x31370000)
    pts = [O.scalar_mul(rng.fr(), O.G1) for _ in range(16)] + [O.INF, O.G1, O.neg(O.G1)]
    return {"note": "seed 0x31370000: 16 random multiples of G, identity, G, -G",
            "compressed": b"".join(O.compress(p) for p in pts).hex(),
            "affine": b"".join(O.aff_to_bytes(p) for p in pts).hex()}


def main():
    files = {"field_kats.json": field_kats(), "point_kats.json": point_kats(), "msm_kats.json": msm_kats(),
             "schema_kats.json": schema_kats(), "wire_kats.json": wire_kats()}
    for name, data in files.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, os.path.getsize(os.path.join(HERE, name)))


if __name__ == "__main__":
    main()
