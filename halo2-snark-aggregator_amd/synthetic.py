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

    __slots__ = ("key", "keys", "commitments", "evals", "rotations", "points", "w", "v", "u", "_c_keys", "_c_rot")

    def __init__(self, key, keys, commitments, evals, rotations, points, w, v, u):
        self.key, self.keys, self.commitments, self.evals = key, keys, commitments, evals
        self.rotations, self.points, self.w, self.v, self.u = rotations, points, w, v, u
        self._c_keys = self._c_rot = None       # C arrays of the keys / rotations, made on first use (marshalling only)

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
    if spec._c_keys is None:
        import ctypes as C
        spec._c_keys = builder.keys_array(spec.keys)
        spec._c_rot = (C.c_int32 * len(spec.rotations))(*spec.rotations)
    qnodes = builder.evaluation_queries(spec._c_keys, spec.commitments, spec.evals, wrap=False)
    w_x, w_g = builder.batch_multi_open(spec.key, spec._c_rot, spec.points, qnodes, spec.w, spec.v, spec.u)
    return MultiOpenProof(w_x, w_g), qnodes[0]


# ---------------------------------------------------------------------------------------------------------------------
# A synthetic verifying-key shape + well-formed random transcripts for the WHOLE pipeline (h2agg_verify_aggregation):
# transcript replay (Poseidon), expressions, queries, fold, both multi_exps, pairing.  The transcripts are not proofs of
# anything (no prover here): every byte decodes, every challenge is derived, the pairing check runs and rejects.
class CircuitShape:
    """the fields of the "H2VK" description (include/h2agg.h); names follow halo2's accessors"""

    def __init__(self, k: int, n_advice: int, pool: Sequence[bytes], seed: int = 0x5AFE, n_lookups: int = 1):
        fr = fr_stream(seed)
        self.k = k
        self.num_advice_columns = n_advice
        self.num_instance_columns = 1
        self.num_challenges = 0
        self.advice_column_phase = [0] * n_advice
        self.challenge_phase = []
        self.advice_queries = [(c, 0) for c in range(n_advice)] + [(c, 1) for c in range(0, n_advice, 7)] + [(0, -1)]
        self.instance_queries = [(0, 0)]
        self.fixed_queries = [(0, 0), (1, 0)]
        self.degree = 5                                                     # chunks of 3 permutation columns
        self.blinding_factors = 5
        self.permutation_columns = [("advice", c) for c in range(7)] + [("fixed", 0), ("instance", 0)]   # 3 sets
        self.fixed_commitments = [pool[1], pool[2]]
        self.permutation_commitments = [pool[3 + i] for i in range(len(self.permutation_columns))]
        self.vk_scalar = int.from_bytes(fr(), "little")
        A = lambda i: ("advice", i)
        F = lambda i: ("fixed", i)
        # a few gates in the style of the sample circuits: q * (a * b - c), q * (a + b - c), ...
        self.gates = [[("product", F(0), ("sum", ("product", A(3 * g), A(3 * g + 1)), ("neg", A(3 * g + 2))))]
                      for g in range(min(8, n_advice // 3))]
        self.gates.append([("product", F(1), ("sum", ("scaled", A(0), 7), ("neg", ("instance", 0))))])
        self.lookups = [([A(j), ("product", F(0), A(j + 1))], [F(1), A(j + 2)]) for j in range(n_lookups)]
        self.n_sets = (len(self.permutation_columns) + 2) // 3

    def rotations(self):
        rots = []
        for r in ([0] + [r for _c, r in self.advice_queries] + [1, -(self.blinding_factors + 1)] +
                  ([0, -1, 1] if self.lookups else [])):
            if r not in rots:
                rots.append(r)
        return rots

    def proof_items(self):
        """(number of points before the evaluations, number of evaluations, number of W points)"""
        n_pts = self.num_advice_columns + 2 * len(self.lookups) + self.n_sets + len(self.lookups) + 1 + (self.degree - 1)
        n_evals = (len(self.instance_queries) + len(self.advice_queries) + len(self.fixed_queries) + 1 +
                   len(self.permutation_commitments) + (3 * self.n_sets - 1) + 5 * len(self.lookups))
        return n_pts, n_evals, len(self.rotations())

    def random_transcript(self, pool_compressed: Sequence[bytes], seed: int) -> bytes:
        fr = fr_stream(seed)
        n_pts, n_evals, n_w = self.proof_items()
        npool = len(pool_compressed)
        pts = b"".join(pool_compressed[(seed * 31 + j) % npool] for j in range(n_pts))
        evals = b"".join(fr() for _ in range(n_evals))
        w = b"".join(pool_compressed[(seed * 17 + 5 + j) % npool] for j in range(n_w))
        return pts + evals + w
