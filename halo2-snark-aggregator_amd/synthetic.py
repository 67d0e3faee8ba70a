"""Synthetic, shape-faithful proof data for the aggregation path (bench.py's `aggregate` leg and the config-sized tests).

Real halo2 proofs cannot be produced in this image (no Rust prover), so a "proof" here is what the transcript reader and
`VerifierParams::queries` hand to the path (halo2-snark-aggregator-api/src/systems/halo2/params.rs:156-223): a list of
(rotation, commitment key, evaluation point, commitment, evaluation) queries in the reference's order, the W commitments
(one per rotation group, multiopen.rs:45-48) and the challenges v, u.  The shape follows SURVEY.md 8(d) config 3: one
instance column, `n_advice` advice columns opened at x, every 7th also at omega*x, 3 permutation-product commitments at
omega^-(blinding+1)*x  ->  P = 1 + n_advice + ceil(n_advice / 7) + 3 queries, 3 rotation groups.

Everything is derived from (seed, global proof index), so every rank builds the same proofs.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class ProofSpec:
    """byte-level proof data: keys[i], commitments (64 B each), evals (32 B each), rotations[i], points (32 B each),
    w (64 B each), v, u (32 B)"""

    __slots__ = ("key", "keys", "commitments", "evals", "rotations", "points", "w", "v", "u")

    def __init__(self, key, keys, commitments, evals, rotations, points, w, v, u):
        self.key, self.keys, self.commitments, self.evals = key, keys, commitments, evals
        self.rotations, self.points, self.w, self.v, self.u = rotations, points, w, v, u

    @property
    def nq(self) -> int:
        return len(self.keys)


def fr_stream(seed: int):
    """uniform Fr elements as 32-byte little-endian strings: 512-bit draws reduced mod r (the from_bytes_wide rule,
    mock/transcript_encode.rs:14-21)"""
    rng = np.random.Generator(np.random.PCG64(seed))

    def fr() -> bytes:
        return (int.from_bytes(rng.bytes(64), "little") % R_MOD).to_bytes(32, "little")
    return fr


def point_pool(eng, seed: int, n: int = 256) -> List[bytes]:
    """n valid affine points k*G from the scalar-mul kernel (commitment VALUES do not change the cost of the path)"""
    fr = fr_stream(seed)
    g_aff = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    aff = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(g_aff * n, b"".join(fr() for _ in range(n))))
    return [aff[64 * i:64 * i + 64] for i in range(n)]


def make_proofs(pool: Sequence[bytes], n_total: int, n_advice: int, seed: int = 0xA66) -> Tuple[List[ProofSpec], bytes]:
    """-> (proof specs for global indices 0 .. n_total-1, aggregation challenge lambda)"""
    fr = fr_stream(seed)
    lam = fr()
    npool = len(pool)
    specs = []
    for i in range(n_total):
        x, xw, xl = fr(), fr(), fr()
        qs = [(0, "p%d_instance_commitments0" % i, x)]
        qs += [(0, "p%d_advice_commitments%d" % (i, c), x) for c in range(n_advice)]
        qs += [(1, "p%d_advice_commitments%d" % (i, c), xw) for c in range(0, n_advice, 7)]
        qs += [(-6, "p%d_perm%d" % (i, c), xl) for c in range(3)]
        commitments = b"".join(pool[(i * 131 + k) % npool] for k in range(len(qs)))
        evals = b"".join(fr() for _ in qs)
        w = b"".join(pool[(i + 1 + j) % npool] for j in range(3))
        v, u = fr(), fr()
        specs.append(ProofSpec("p%d" % i, [k for _r, k, _z in qs], commitments, evals, [r for r, _k, _z in qs],
                               b"".join(z for _r, _k, z in qs), w, v, u))
    return specs, lam


def build_proof(builder, MultiOpenProof, spec: ProofSpec):
    """n x EvaluationQuery::new + batch_multi_open_proofs in the C++ host layer -> (MultiOpenProof, first query node)"""
    qnodes = builder.evaluation_queries(spec.keys, spec.commitments, spec.evals, wrap=False)
    w_x, w_g = builder.batch_multi_open(spec.key, spec.rotations, spec.points, qnodes, spec.w, spec.v, spec.u)
    return MultiOpenProof(w_x, w_g), qnodes[0]
