"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Pure-Python restatement of the verifier-schema layer that sits directly above the
arithmetic chips on the hot path:

  * EvaluationQuerySchema + eval / eval_prepare / estimate
        halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:15-39, 62-84, 172-330
  * the GWC multi-open fold (Horner in v per rotation group, Horner in u over groups)
        halo2-snark-aggregator-api/src/systems/halo2/multiopen.rs:23-102
  * the aggregation fold  acc = acc * lambda + proof
        halo2-snark-aggregator-api/src/systems/halo2/verify.rs:926-938
  * the tail of evaluate_multiopen_proof  (left = w_x + e*G, right = w_g - e*G)
        halo2-snark-aggregator-api/src/systems/halo2/verify.rs:705-731

Parity pinning: see oracle/bn254.py header ("parity unpinned" by reference vectors; the
reference has none).  Points are affine tuples / None, scalars are ints mod r.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

from . import bn254 as O


# ----------------------------------------------------------------------------- chips
class OracleCtx:
    """MockChipCtx (mock/arith/field.rs:11-21)."""

    def __init__(self):
        self.point_list: List[str] = []
        self.tag: str = ""

    def __str__(self):
        return "(total points: %d)" % len(self.point_list)


class OracleFieldChip:
    """MockFieldChip (mock/arith/field.rs:23-146) + ArithFieldChip defaults (arith/field.rs:37-104)."""

    def add(self, ctx, a, b):
        return (a + b) % O.R

    def sub(self, ctx, a, b):
        return (a - b) % O.R

    def mul(self, ctx, a, b):
        return a * b % O.R

    def div(self, ctx, a, b):
        return a * O.inv(b, O.R) % O.R

    def square(self, ctx, a):
        return a * a % O.R

    def assign_zero(self, ctx):
        return 0

    def assign_one(self, ctx):
        return 1

    def assign_const(self, ctx, c):
        return c % O.R

    assign_var = assign_const

    def to_value(self, v):
        return v

    def normalize(self, ctx, v):
        return v

    def sum_with_coeff_and_constant(self, ctx, a_with_coeff, b):
        acc = b
        for x, coeff in a_with_coeff:
            acc = (acc + x * coeff) % O.R
        return acc

    def sum_with_constant(self, ctx, a, b):
        return self.sum_with_coeff_and_constant(ctx, [(x, 1) for x in a], b)

    def mul_add_constant(self, ctx, a, b, c):
        return (a * b + c) % O.R

    def mul_add(self, ctx, a, b, c):
        return self.add(ctx, self.mul(ctx, a, b), c)

    def mul_add_accumulate(self, ctx, a, b):
        acc = self.assign_zero(ctx)
        for v in a:
            acc = self.mul_add(ctx, acc, b, v)
        return acc

    def pow_constant(self, ctx, base, exponent: int):
        assert exponent >= 1
        acc = base
        second_bit = 1
        while second_bit <= exponent:
            second_bit <<= 1
        second_bit >>= 2
        while second_bit > 0:
            acc = self.square(ctx, acc)
            if exponent & second_bit:
                acc = self.mul(ctx, acc, base)
            second_bit >>= 1
        return acc


class OracleEccChip:
    """MockEccChip (mock/arith/ecc.rs:8-130)."""

    def add(self, ctx, a, b):
        return O.add(a, b)

    def sub(self, ctx, a, b):
        return O.sub(a, b)

    def assign_zero(self, ctx):
        return O.INF

    def assign_one(self, ctx):
        return O.G1

    def assign_const(self, ctx, c):
        return c

    assign_var = assign_const

    def to_value(self, v):
        return v

    def normalize(self, ctx, v):
        return v

    def scalar_mul(self, ctx, lhs, rhs):
        return O.scalar_mul(lhs, rhs)

    def scalar_mul_constant(self, ctx, lhs, rhs):
        return O.scalar_mul(lhs, rhs)

    def multi_exp(self, ctx, points, scalars):
        ctx.point_list = [O.debug_fmt(p) for p in points]       # mock/arith/ecc.rs:112-116
        return O.multi_exp(points, scalars)


# ----------------------------------------------------------------------------- schema AST
@dataclass
class CommitQuery:                 # evaluation.rs:7-12
    key: str
    commitment: object = None      # Option<P>
    eval: Optional[int] = None     # Option<S>


class Schema:
    """EvaluationQuerySchema (evaluation.rs:15-27).  kind in {commitment, eval, scalar, add, mul}."""

    __slots__ = ("kind", "cq", "s", "l", "r", "lh", "rh")

    def __init__(self, kind, cq=None, s=None, l=None, r=None):
        self.kind, self.cq, self.s, self.l, self.r = kind, cq, s, l, r
        # cached has-commitment flags of the children (the `bool` in each Box<(_, bool)>)
        self.lh = l.has_commitment() if l is not None else False
        self.rh = r.has_commitment() if r is not None else False

    def has_commitment(self) -> bool:          # evaluation.rs:30-38
        if self.kind == "commitment":
            return True
        if self.kind in ("eval", "scalar"):
            return False
        return self.lh or self.rh

    def __add__(self, other):                  # evaluation.rs:62-72
        return Schema("add", l=self, r=other)

    def __mul__(self, other):                  # evaluation.rs:74-84
        return Schema("mul", l=self, r=other)

    # ---- eval_prepare (evaluation.rs:205-293)
    def eval_prepare(self, ctx, schip, one, scalar):
        k = self.kind
        if k == "commitment":
            return [(self.cq.key, self.cq.commitment, scalar)]
        if k == "eval":
            e = schip.mul(ctx, scalar, self.cq.eval) if scalar is not None else self.cq.eval
            return [("", None, e)]
        if k == "scalar":
            s = schip.mul(ctx, self.s, scalar) if scalar is not None else self.s
            return [("", None, s)]
        if k == "add":
            if not self.lh and not self.rh:
                l = self.l.eval_prepare(ctx, schip, one, None)
                r = self.r.eval_prepare(ctx, schip, one, None)
                assert len(l) == 1 and len(r) == 1
                total = schip.add(ctx, l[0][2], r[0][2])
                if scalar is not None:
                    total = schip.mul(ctx, scalar, total)
                return [("", None, total)]
            res = []
            for side in (self.l, self.r):
                for ev in side.eval_prepare(ctx, schip, one, scalar):
                    found = None
                    for i, p in enumerate(res):
                        if p[0] == ev[0]:
                            found = i
                            break
                    if found is not None:
                        p = res[found]
                        a = p[2] if p[2] is not None else one
                        b = ev[2] if ev[2] is not None else one
                        res[found] = (p[0], p[1], schip.add(ctx, a, b))
                    else:
                        res.append(ev)
            return res
        if k == "mul":
            if not self.lh:
                s = self.l.eval_prepare(ctx, schip, one, None)
                rem = self.r
            else:
                s = self.r.eval_prepare(ctx, schip, one, None)
                rem = self.l
            assert len(s) == 1
            s = s[0][2]
            if scalar is not None:
                s = schip.mul(ctx, scalar, s)
            return rem.eval_prepare(ctx, schip, one, s)
        raise AssertionError(k)

    # ---- eval (evaluation.rs:172-203)
    def eval(self, ctx, schip, pchip, one):
        points = self.eval_prepare(ctx, schip, one, None)
        point_names = [name for (name, _p, _s) in points]
        s = None
        for b in points:
            if b[0] == "":
                s = b[2]
                break
        p_wo_scalar = [b[1] for b in points if b[2] is None and b[1] is not None]
        # NB the reference filters with `p.and_then(...)`, and represents "no point" as None; the
        # identity *is* a point.  Here points are tuples or O.INF (= None) so presence is tracked
        # by the key: only the "" entry has no point.
        pl, sl = [], []
        for (name, p, sc) in points:
            if name != "" and sc is not None:
                pl.append(p)
                sl.append(sc)
        acc = pchip.multi_exp(ctx, pl, sl)
        for p in p_wo_scalar:
            acc = pchip.add(ctx, acc, p)
        return acc, s, point_names

    # ---- estimate (evaluation.rs:295-330)
    def estimate(self, scalar: bool = False) -> int:
        k = self.kind
        if k == "commitment":
            return 1
        if k in ("eval", "scalar"):
            return 1 if scalar else 0
        if k == "add":
            if not self.lh and not self.rh:
                n = self.l.estimate(False) + self.r.estimate(False)
                return n + 1 if scalar else n
            return self.l.estimate(scalar) + self.r.estimate(scalar)
        if k == "mul":
            return self.r.estimate(True) if not self.lh else self.l.estimate(True)
        raise AssertionError(k)


def commit(cq: CommitQuery) -> Schema:     # commit! macro  evaluation.rs:41-46
    return Schema("commitment", cq=cq)


def evalq(cq: CommitQuery) -> Schema:      # eval! macro    evaluation.rs:48-53
    return Schema("eval", cq=cq)


def scalar(s: int) -> Schema:              # scalar! macro  evaluation.rs:55-60
    return Schema("scalar", s=s)


def evaluation_query(rotation: int, key: str, point: int, commitment, ev: int):
    """EvaluationQuery::new (evaluation.rs:100-118): schema = [C] + eval."""
    cq = CommitQuery(key, commitment, ev)
    return (rotation, point, commit(cq) + evalq(cq))


# ----------------------------------------------------------------------------- multiopen fold
@dataclass
class MultiOpenProof:              # multiopen.rs:10-13
    w_x: Schema
    w_g: Schema

    def __str__(self):             # multiopen.rs:15-20
        return "(estimated scalar mult of points: %d)" % (self.w_x.estimate() + self.w_g.estimate())


def batch_multi_open_proofs(key: str, queries, w: list, v: int, u: int) -> MultiOpenProof:
    """VerifierParams::get_point_schemas + batch_multi_open_proofs (multiopen.rs:23-102).
    `queries` = list of (rotation, point, schema) in VerifierParams::queries order."""
    points = []                                     # [(rotation, point, [schemas])], first-seen order :31-43
    for rot, pt, s in queries:
        for g in points:
            if g[0] == rot:
                g[2].append(s)
                break
        else:
            points.append((rot, pt, [s]))
    assert len(w) == len(points)                    # :48
    proofs = []
    for i, (_rot, pt, schemas) in enumerate(points):
        acc = None
        for q in reversed(schemas):                 # .rev().reduce(|acc,q| scalar!(v)*acc + q)  :56-60
            acc = q if acc is None else scalar(v) * acc + q
        proofs.append((pt, acc, w[i]))
    w_x = w_g = None
    for i in range(len(proofs) - 1, -1, -1):        # .enumerate().rev()  :82
        pt, s, wi = proofs[i]
        wq = CommitQuery("%s_w%d" % (key, i), wi, None)
        w_x = commit(wq) if w_x is None else scalar(u) * w_x + commit(wq)
        if w_g is None:
            w_g = scalar(pt) * commit(wq) + s
        else:
            w_g = scalar(u) * w_g + scalar(pt) * commit(wq) + s
    return MultiOpenProof(w_x, w_g)


def aggregate_fold(proofs: List[MultiOpenProof], lam: int) -> MultiOpenProof:
    """verify.rs:926-938."""
    acc = None
    for p in proofs:
        if acc is None:
            acc = p
        else:
            acc = MultiOpenProof(acc.w_x * scalar(lam) + p.w_x, acc.w_g * scalar(lam) + p.w_g)
    return acc


def evaluate_multiopen_proof(ctx, schip, pchip, proof: MultiOpenProof):
    """verify.rs:705-731 (pairing check :733-740 is print-only in the reference and out of scope)."""
    one = schip.assign_one(ctx)
    left_s, left_e, names_x = proof.w_x.eval(ctx, schip, pchip, one)
    right_s, right_e, names_g = proof.w_g.eval(ctx, schip, pchip, one)
    gen = pchip.assign_one(ctx)
    left = left_s if left_e is None else pchip.add(ctx, left_s, pchip.scalar_mul(ctx, left_e, gen))
    right = right_s if right_e is None else pchip.sub(ctx, right_s, pchip.scalar_mul(ctx, right_e, gen))
    return pchip.to_value(left), pchip.to_value(right), names_x + names_g


def final_pair_bytes(left, right, instances=()) -> bytes:
    """`verify_circuit_final_pair.data` layout (halo2-snark-aggregator-circuit/src/fs.rs:182-195)."""
    out = O.aff_to_bytes(left) + O.aff_to_bytes(right)
    for s in instances:
        out += O.fe_to_bytes(s)
    return out


def final_pair_to_instances(left, right, instances=()):
    """final_pair_to_instances (halo2-snark-aggregator-circuit/src/verify_circuit.rs:768-804) restated with
    field arithmetic, as the reference does: limbs via % and >>, combined with limb_modulus_exps in Fr."""
    lm = 1 << 68
    exps = [pow(lm, k, O.R) for k in range(4)]                       # limb_modulus_exps (integer_chip.rs:105-110)

    def limbs(v):
        out = []
        for _ in range(3):
            out.append(v % lm)
            v >>= 68
        out.append(v)
        return [x % O.R for x in out]
    res = []
    for pt in (left, right):
        x, y = limbs(pt[0]), limbs(pt[1])
        last = exps[2] if (y[0] & 1) else 0
        res.append((x[0] * exps[0] + x[1] * exps[1]) % O.R)
        res.append((x[2] * exps[0] + x[3] * exps[1] + last) % O.R)
    return res + [s % O.R for s in instances]
