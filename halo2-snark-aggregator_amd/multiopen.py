"""GWC multi-open batching and the aggregation fold, written against a schema *builder* (the GPU backend's
`SchemaBuilder`, or any object with the same `commit / evalq / scalar` constructors whose nodes support
`+` and `*`).  Pure AST construction — no field or group arithmetic happens here.

Status: the PRODUCT implementation of this fold is C++ (`Schema::batch_multi_open(_regs)` in csrc/schema.hpp, reached
through h2agg_schema_batch_multi_open and h2agg_verify_aggregation).  This Python twin is what the tests and bench.py's
`aggregate` leg build their trees with, node by node through the C ABI — a second route to the same tree, kept for
differential checks (tests/test_gpu_schema.py::test_cpp_batch_builders_match_oracle), not a second product path.

Mirrors
  EvaluationQuery::new                          halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:100-118
  VerifierParams::get_point_schemas             .../multiopen.rs:23-69
  VerifierParams::batch_multi_open_proofs       .../multiopen.rs:71-102
  the fold in verify_aggregation_proofs_in_chip .../verify.rs:926-938
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Sequence, Tuple


@dataclass
class EvaluationQuery:            # evaluation.rs:92-97
    rotation: int
    point: bytes                  # the evaluation point z (Fr, 32-byte LE canonical)
    s: Any                        # schema node


@dataclass
class MultiOpenProof:             # multiopen.rs:10-13
    w_x: Any
    w_g: Any

    def __str__(self):            # multiopen.rs:15-20
        return "(estimated scalar mult of points: %d)" % (self.w_x.estimate() + self.w_g.estimate())


def evaluation_query(b, CommitQuery, rotation: int, key: str, point: bytes, commitment: bytes, ev: bytes):
    """EvaluationQuery::new: schema = [C] + eval."""
    cq = CommitQuery(key, commitment, ev)
    return EvaluationQuery(rotation, point, b.commit(cq) + b.evalq(cq))


def batch_multi_open_proofs(b, CommitQuery, key: str, queries: Sequence[EvaluationQuery], w: Sequence[bytes],
                            v: bytes, u: bytes) -> MultiOpenProof:
    """get_point_schemas + batch_multi_open_proofs.  `w` = the W commitments (affine bytes), one per
    rotation group in first-seen order (multiopen.rs:45-48 asserts the counts match)."""
    groups: List[Tuple[int, bytes, list]] = []
    for q in queries:                                           # :33-43 group by rotation, first-seen order
        for g in groups:
            if g[0] == q.rotation:
                g[2].append(q.s)
                break
        else:
            groups.append((q.rotation, q.point, [q.s]))
    assert len(w) == len(groups), "assert_eq!(self.w.len(), points.len()) (multiopen.rs:48)"
    proofs = []
    for i, (_rot, point, schemas) in enumerate(groups):
        acc = None
        for q in reversed(schemas):                             # .rev().reduce(|acc, q| scalar!(v) * acc + q)  :56-60
            acc = q if acc is None else b.scalar(v) * acc + q
        proofs.append((point, acc, w[i]))
    w_x = w_g = None
    for i in range(len(proofs) - 1, -1, -1):                    # .enumerate().rev()  :82
        point, s, wi = proofs[i]
        wq = CommitQuery("%s_w%d" % (key, i), wi, None)
        w_x = b.commit(wq) if w_x is None else b.scalar(u) * w_x + b.commit(wq)
        if w_g is None:
            w_g = b.scalar(point) * b.commit(wq) + s
        else:
            w_g = b.scalar(u) * w_g + b.scalar(point) * b.commit(wq) + s
    return MultiOpenProof(w_x, w_g)


def aggregate_fold(b, proofs: Sequence[MultiOpenProof], lam: bytes) -> MultiOpenProof:
    """acc = acc * lambda + proof (verify.rs:926-938): proof i ends up weighted by lambda^(N-1-i)."""
    acc = None
    for p in proofs:
        if acc is None:
            acc = p
        else:
            acc = MultiOpenProof(acc.w_x * b.scalar(lam) + p.w_x, acc.w_g * b.scalar(lam) + p.w_g)
    return acc
