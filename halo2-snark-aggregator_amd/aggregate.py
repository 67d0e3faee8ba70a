"""Sharded aggregation of N proofs across ranks (one process per GPU).

The reference folds all proofs into one schema and evaluates it on one CPU thread
(verify_aggregation_proofs_in_chip, halo2-snark-aggregator-api/src/systems/halo2/verify.rs:835-942).  The
fold is linear: (W_x, W_g) = sum_i lambda^(N-1-i) * (W_x_i, W_g_i), so the proofs shard across ranks:

    rank r takes proofs r, r+G, r+2G, ...            (round-robin, no data-path collective)
    builds its local schema  sum_j lambda^(N-1-i_j) * proof_{i_j}   (powers of lambda are schema nodes:
        they are evaluated on the device tape, the host does no field arithmetic)
    evaluates it on its GPU -> partial (left, right) affine points (e*G terms included: linear too)
    ONE all-gather of 2 x 64 bytes per rank, then every rank sums the G partials locally.

RCCL has no reduction over group elements (ncclRedOp_t is sum/prod/min/max on numeric dtypes), so
BASELINE.json's "all-reduce of (W_x, W_e)" is an all-gather + local EC adds.

`backend` is duck-typed so the distributed logic can be tested on CPU with gloo (tests inject an
oracle-backed backend); the product backend is `GpuBackend` below and has no fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from .multiopen import MultiOpenProof

IDENTITY_AFF = bytes(64)


def shard_indices(n: int, world: int, rank: int) -> List[int]:
    return list(range(rank, n, world))


def lambda_power(b, lam: bytes, e: int):
    """lambda^e as a schema node (e >= 1), square-and-multiply over Scalar nodes (Mul of two
    commitment-free sides is a scalar product in eval_prepare, evaluation.rs:271-291)."""
    assert e >= 1
    result = None
    base = b.scalar(lam)
    while e:
        if e & 1:
            result = base if result is None else result * base
        e >>= 1
        if e:
            base = base * base
    return result


def local_weighted_proof(b, proofs: Sequence[MultiOpenProof], indices: Sequence[int], n_total: int,
                         lam: bytes) -> Optional[MultiOpenProof]:
    """sum_j lambda^(N-1-i_j) * proofs[j] for this rank's (ascending) global indices."""
    acc = None
    for j, (p, i) in enumerate(zip(proofs, indices)):
        if acc is None:
            acc = p
        else:
            gap = i - indices[j - 1]
            lg = lambda_power(b, lam, gap)
            acc = MultiOpenProof(acc.w_x * lg + p.w_x, acc.w_g * lg + p.w_g)
    if acc is None:
        return None
    tail = n_total - 1 - indices[-1]
    if tail > 0:
        lt = lambda_power(b, lam, tail)
        acc = MultiOpenProof(acc.w_x * lt, acc.w_g * lt)
    return acc


def slice_bounds(n: int, world: int, rank: int):
    """contiguous slice [lo, hi) of n points for this rank (the remainder goes to the first ranks)"""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def msm_sliced(backend, bases_aff: bytes, scalars: bytes, dist=None, device=None) -> bytes:
    """One multi_exp split across the ranks by POINTS (second grain): rank r runs a full Pippenger on
    its contiguous slice of (bases, scalars), the 64-byte partial results are all-gathered and every rank adds them —
    no other communication.  `bases_aff` / `scalars` are this call's full inputs on every rank (a caller that keeps the
    bases resident per rank passes only its slice and world = 1 semantics through `backend.msm`).  Returns the affine
    result, identical on every rank.  An empty slice contributes the identity."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    n = len(scalars) // 32
    lo, hi = slice_bounds(n, world, rank)
    part = backend.msm(bases_aff[64 * lo:64 * hi], scalars[32 * lo:32 * hi]) if hi > lo else IDENTITY_AFF
    if dist is None:
        return part
    import torch
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return backend.sum_affine([bytes(g.cpu().numpy().tobytes()) for g in gathered])


class GpuBackend:
    """Product backend: everything runs through libh2agg.so."""

    def __init__(self, pkg, eng):
        self.pkg, self.eng = pkg, eng
        self.CommitQuery = pkg.CommitQuery

    def new_builder(self):
        return self.pkg.SchemaBuilder(self.eng)

    def prepare(self, b, proof: MultiOpenProof):
        """host half of evaluate() ahead of time (h2agg_evaluate_multiopen_prepare): no device work, no wait"""
        b.evaluate_multiopen_prepare(proof.w_x, proof.w_g)

    def evaluate(self, b, proof: MultiOpenProof):
        left, right, _names = b.evaluate_multiopen_proof(proof.w_x, proof.w_g)
        return left, right

    def msm(self, bases_aff: bytes, scalars: bytes) -> bytes:
        """h2agg_g1_msm + to_affine (MockEccChip::multi_exp, mock/arith/ecc.rs:106-129)"""
        return self.eng.g1_batch_to_affine(self.eng.g1_msm(bases_aff, scalars))

    def sum_affine(self, pts: Sequence[bytes]) -> bytes:
        one = (1).to_bytes(32, "little")
        jac = b"".join((p + one) if p != IDENTITY_AFF else (bytes(32) + one + bytes(32)) for p in pts)
        return self.eng.g1_batch_to_affine(self.eng.g1_sum(jac))


def aggregate_sharded(backend, build_local_proofs, n_total: int, lam: bytes, dist=None, device=None, comm=None,
                      rank_world=None):
    """Returns the final pair (left_aff, right_aff), identical on every rank.

    build_local_proofs(builder, indices) -> list of MultiOpenProof for those global proof indices, or (list, finish):
          `finish()` is then called once the fold is built and the evaluation's host half is done (backend.prepare),
          right before the evaluation — the place to wait for commitments the device is still computing and to patch
          them in (query_set_commitment).
    dist: torch.distributed (initialised) or None for a single process.
    comm: an H2Agg engine holding an RCCL communicator (comm_init_rank): the exchange then happens INSIDE the C ABI
          (h2agg_allgather_add_points: all-gather over RCCL + local EC adds) — what a non-Python host binds; rank / world
          come from `rank_world` or, if that is None, from the communicator / dist."""
    if comm is not None:
        world = comm.comm_size()
        if world < 1:
            raise ValueError("aggregate_sharded: `comm` holds no communicator (comm_init_rank was not called)")
        # the communicator knows its own rank (h2agg_comm_rank); an explicit rank_world / dist must agree with it — every
        # process silently taking shard 0 would sum `world` copies of it into a wrong pair
        rank = comm.comm_rank()
        claimed = rank_world[0] if rank_world is not None else (dist.get_rank() if dist is not None else rank)
        if rank < 0 or claimed != rank or (rank_world is not None and rank_world[1] != world):
            raise ValueError("aggregate_sharded: rank/world (%r) disagree with the communicator's (%d, %d)"
                             % (rank_world if rank_world is not None else claimed, rank, world))
    else:
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
    idx = shard_indices(n_total, world, rank)
    b = backend.new_builder()
    built = build_local_proofs(b, idx)
    proofs, finish = built if isinstance(built, tuple) else (built, None)
    local = local_weighted_proof(b, proofs, idx, n_total, lam)
    if local is not None and finish is not None and hasattr(backend, "prepare"):
        backend.prepare(b, local)
    if finish is not None:
        finish()
    if local is None:
        left, right = IDENTITY_AFF, IDENTITY_AFF
    else:
        left, right = backend.evaluate(b, local)
    if comm is not None:
        one = (1).to_bytes(32, "little")
        jac = b"".join((p + one) if p != IDENTITY_AFF else (bytes(32) + one + bytes(32)) for p in (left, right))
        out = comm.allgather_add_points(jac)                          # the only collective: 192 B per rank
        return out[:64], out[64:]
    if dist is None:
        return left, right
    import torch
    mine = torch.frombuffer(bytearray(left + right), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)                                   # the only collective: 128 B per rank
    parts = [bytes(g.cpu().numpy().tobytes()) for g in gathered]
    return (backend.sum_affine([p[:64] for p in parts]), backend.sum_affine([p[64:] for p in parts]))


def aggregate_sharded_from_bytes(verifier_mod, eng, circuits_local, global_index, n_total: int, rank: int, world: int,
                                 dist=None, device=None, s_g2=None, g2=None):
    """The sharded aggregation FROM PROOF BYTES: no challenge is passed in.  Each rank hands its proofs' transcripts to
    h2agg_verify_aggregation_sharded, which replays them, exchanges every proof's last squeeze (32 B) so that every rank
    derives the SAME lambda (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:909-913, :924), folds its proofs with the
    powers lambda^(N-1-i) (:926-938), evaluates its partial pair on its GPU and exchanges + sums the partials.

    circuits_local / global_index: as verifier.verify_aggregation_sharded.  dist: an initialised torch.distributed group used
    as the transport (gloo, or nccl = RCCL with `device`); None: the engine's own RCCL communicator (eng.comm_init_rank), the
    exchange then never leaves the C ABI.  Returns (left_aff, right_aff, lambda, pairing_ok or None), equal on every rank."""
    allgather = verifier_mod.dist_allgather(dist, device) if dist is not None else None
    return verifier_mod.verify_aggregation_sharded(eng, circuits_local, global_index, n_total, rank, world, allgather, s_g2, g2)
