"""Test-side TRAPDOOR prover: produces transcripts that the halo2/KZG verifier of the reference ACCEPTS, for arbitrary
constraint-system shapes, without a halo2 prover (there is no Rust toolchain in this image).

How: the "trusted setup" secret tau is known here.  Every commitment is c*G for a scalar c the prover picks, every
evaluation is a scalar it picks; the only relation the verifier ever checks is the final pairing
    e(sum u^i W_i, [tau]_2) = e(sum u^i (z_i W_i + sum_k v^k (C_ik - e_ik G)), [1]_2)            (multiopen.rs:71-102, verify.rs:733-739)
which holds iff W_i = (sum_k v^k (c_ik - e_ik)) / (tau - z_i) * G — computable in the exponent when tau is known.  (A real
prover needs (C, e) to be a polynomial and its evaluation; with tau every pair is "consistent".)  The gate / permutation /
lookup identities enter through the vanishing query, whose expected evaluation the prover computes with the SAME oracle
code the verifier restatement uses (oracle/verifier.py `queries`), so a verifier implementation that evaluates them
differently from the oracle is rejected by the pairing.

This is checker-side infrastructure: it lives under tests/ and uses oracle/.
"""
from __future__ import annotations

from typing import List

from oracle import bn254 as O
from oracle import pairing as E
from oracle import poseidon as P
from oracle import schema as S
from oracle import verifier as V

R = O.R


class Setup:
    """ParamsKZG stand-in with a known secret: g_lagrange[i] = L_i(tau) * G, s_g2 = tau * G2"""

    def __init__(self, k: int, tau: int, n_lagrange: int):
        self.k, self.tau = k, tau % R
        n = 1 << k
        omega = V.omega_for_k(k)
        tn = (pow(self.tau, n, R) - 1) % R
        ninv = O.inv(n % R, R)
        self.lagrange_dlogs = []
        wi = 1
        for _ in range(n_lagrange):
            self.lagrange_dlogs.append(wi * tn % R * ninv % R * O.inv((self.tau - wi) % R, R) % R)
            wi = wi * omega % R
        self.g_lagrange = [O.scalar_mul(d, O.G1) for d in self.lagrange_dlogs]
        self.s_g2 = E.g2_mul(self.tau, E.G2)
        self.g2 = E.G2


class DlogChip(S.OracleEccChip):
    """MockEccChip "in the exponent": values are discrete logs; real points are translated through `dmap`"""

    def __init__(self, dmap):
        self.dmap = dmap

    def _d(self, p):
        if isinstance(p, int):
            return p % R
        if p is O.INF:
            return 0
        return self.dmap[p]

    def add(self, ctx, a, b):
        return (self._d(a) + self._d(b)) % R

    def sub(self, ctx, a, b):
        return (self._d(a) - self._d(b)) % R

    def assign_one(self, ctx):
        return 1

    def assign_zero(self, ctx):
        return 0

    def scalar_mul(self, ctx, lhs, rhs):
        return lhs * self._d(rhs) % R

    scalar_mul_constant = scalar_mul

    def multi_exp(self, ctx, points, scalars):
        assert points, "multi_exp of zero pairs"
        return sum(self._d(p) * s for p, s in zip(points, scalars)) % R


def random_expression(rng, cs_shape, depth: int):
    """a random Expression over the query indices of the shape (no selectors)"""
    nf, na, ni, nc = cs_shape
    leaves = []
    if nf:
        leaves.append(lambda: ("fixed", rng.next() % nf))
    if na:
        leaves.append(lambda: ("advice", rng.next() % na))
    if ni:
        leaves.append(lambda: ("instance", rng.next() % ni))
    if nc:
        leaves.append(lambda: ("challenge", rng.next() % nc))
    leaves.append(lambda: ("const", rng.fr()))
    if depth == 0:
        return leaves[rng.next() % len(leaves)]()
    t = rng.next() % 6
    if t == 0:
        return ("neg", random_expression(rng, cs_shape, depth - 1))
    if t == 1:
        return ("sum", random_expression(rng, cs_shape, depth - 1), random_expression(rng, cs_shape, depth - 1))
    if t in (2, 3):
        return ("product", random_expression(rng, cs_shape, depth - 1), random_expression(rng, cs_shape, depth - 1))
    if t == 4:
        return ("scaled", random_expression(rng, cs_shape, depth - 1), rng.fr())
    return leaves[rng.next() % len(leaves)]()


def make_constraint_system(rng, k=5, n_advice=3, n_fixed=2, n_instance=1, n_gates=2, n_lookups=1, degree=4,
                           n_perm_columns=4, phases=(0,), n_challenges=0, dlogs=None):
    """a random circuit shape in the style of the reference's sample circuits (sdk/examples/simple-example.rs): every
    column is queried at the current rotation, some advice columns also at +1 / -1; commitments of the VK are c*G with
    the c recorded in `dlogs` (point -> scalar)"""
    dlogs = dlogs if dlogs is not None else {}

    def point():
        d = rng.fr()
        p = O.scalar_mul(d, O.G1)
        dlogs[p] = d
        return p
    advice_queries = [(c, 0) for c in range(n_advice)] + [(c, 1) for c in range(0, n_advice, 2)] + [(0, -1)]
    fixed_queries = [(c, 0) for c in range(n_fixed)]
    instance_queries = [(c, 0) for c in range(n_instance)]
    shape = (len(fixed_queries), len(advice_queries), len(instance_queries), n_challenges)
    gates = [[random_expression(rng, shape, 3) for _ in range(1 + g % 2)] for g in range(n_gates)]
    lookups = [([random_expression(rng, shape, 2) for _ in range(2)], [random_expression(rng, shape, 1) for _ in range(2)])
               for _ in range(n_lookups)]
    kinds = [("advice", c) for c in range(n_advice)] + [("fixed", c) for c in range(n_fixed)] + \
            [("instance", c) for c in range(n_instance)]
    perm_cols = kinds[:n_perm_columns]
    advice_phase = [phases[c % len(phases)] for c in range(n_advice)]
    challenge_phase = [phases[c % len(phases)] for c in range(n_challenges)]
    return V.ConstraintSystem(
        k=k, num_advice_columns=n_advice, num_instance_columns=n_instance, num_challenges=n_challenges,
        advice_column_phase=advice_phase, challenge_phase=challenge_phase, advice_queries=advice_queries,
        instance_queries=instance_queries, fixed_queries=fixed_queries, gates=gates, lookups=lookups,
        permutation_columns=perm_cols, degree=degree, blinding_factors=5,
        fixed_commitments=[point() for _ in range(n_fixed)], permutation_commitments=[point() for _ in perm_cols],
        vk_scalar=rng.fr())


def prove(cs: V.ConstraintSystem, setup: Setup, rng, instances, dlogs: dict, key: str = "") -> bytes:
    """-> transcript bytes of ONE proof (instances: [inner proof][column][values]) accepted by the verifier"""
    w = P.PoseidonTranscriptWrite()

    def new_point():
        d = rng.fr()
        p = O.scalar_mul(d, O.G1)
        dlogs[p] = d
        return p
    w.common_scalar(cs.vk_scalar % R)
    inst_commitments = []
    for inst in instances:
        row = []
        for column in inst:
            d = sum(v * setup.lagrange_dlogs[i] for i, v in enumerate(column)) % R
            p = O.scalar_mul(d, O.G1) if column else O.INF
            if p is not O.INF:
                dlogs[p] = d
            row.append(p)
            w.common_point(p)
        inst_commitments.append(row)
    nproofs = len(instances)
    for phase in cs.phases():
        for _ in range(nproofs):
            for ph in cs.advice_column_phase:
                if ph == phase:
                    w.write_point(new_point())
        for ph in cs.challenge_phase:
            if ph == phase:
                w.squeeze_challenge_scalar()
    w.squeeze_challenge_scalar()                                   # theta
    for _ in range(nproofs):
        for _ in cs.lookups:
            w.write_point(new_point())
            w.write_point(new_point())
    w.squeeze_challenge_scalar()                                   # beta
    w.squeeze_challenge_scalar()                                   # gamma
    for _ in range(nproofs):
        for _ in range(cs.num_permutation_sets):
            w.write_point(new_point())
    for _ in range(nproofs):
        for _ in cs.lookups:
            w.write_point(new_point())
    w.write_point(new_point())                                     # random commitment
    w.squeeze_challenge_scalar()                                   # y
    for _ in range(cs.quotient_poly_degree):
        w.write_point(new_point())
    w.squeeze_challenge_scalar()                                   # x
    n_evals = nproofs * (len(cs.instance_queries) + len(cs.advice_queries)) + len(cs.fixed_queries) + 1 + \
        len(cs.permutation_commitments) + nproofs * (3 * cs.num_permutation_sets - 1 if cs.num_permutation_sets else 0) + \
        nproofs * 5 * len(cs.lookups)
    for _ in range(n_evals):
        w.write_scalar(rng.fr())
    v = w.squeeze_challenge_scalar()
    # the verifier's own view of what has been written so far (no W yet): queries with their evaluation points
    rd = P.PoseidonTranscriptRead(w.finalize())
    vp = V.build_params(rd, S.OracleEccChip(), S.OracleCtx(), inst_commitments, cs, key)
    assert vp.v == v and vp.w == []
    chip = DlogChip(dlogs)
    groups = []                                                    # multiopen.rs:33-43: by rotation, first-seen order
    for rot, pt, s in V.queries(vp):
        for g in groups:
            if g[0] == rot:
                g[2].append(s)
                break
        else:
            groups.append([rot, pt, [s]])
    for _rot, z, schemas in groups:
        a, vk = 0, 1
        for s in schemas:                                          # sum_k v^k q_k (multiopen.rs:56-60)
            c = e = 0
            for name, pt_, sc in s.eval_prepare(S.OracleCtx(), S.OracleFieldChip(), 1, None):
                if name == "":
                    e = (e + sc) % R
                else:
                    c = (c + chip._d(pt_) * (1 if sc is None else sc)) % R
            a = (a + vk * (c - e)) % R
            vk = vk * v % R
        wd = a * O.inv((setup.tau - z) % R, R) % R
        p = O.scalar_mul(wd, O.G1)
        dlogs[p] = wd
        w.write_point(p)
    return w.finalize()
