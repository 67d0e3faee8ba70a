"""Proof wire format, input side of the path: what `TranscriptRead::{read_point, read_scalar}` consume
(halo2-snark-aggregator-api/src/systems/halo2/transcript.rs:56-119) — a byte stream of 32-byte compressed G1 points and
32-byte little-endian Fr elements in an order fixed by the verifying key.  The reference decodes one element per call, on
the CPU, interleaved with the Poseidon absorption; here the stream of N proofs is split by a layout and ALL points are
decompressed in one launch (`h2agg_g1_batch_decompress`).  The sponge itself is not part of this module (SURVEY.md 8(f)-2:
its constants come from an unvendored crate and cannot be checked in this image).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class ProofFormatError(ValueError):
    """"invalid field element encoding in proof" / short read (transcript.rs:106-113, read_exact)"""


def split_proof(layout: Sequence[str], data: bytes) -> Tuple[List[bytes], List[bytes]]:
    """layout: 'P' (compressed point) / 'S' (scalar) tokens in transcript order.  Returns the raw 32-byte items."""
    if len(data) != 32 * len(layout):
        raise ProofFormatError("proof has %d bytes, layout needs %d" % (len(data), 32 * len(layout)))
    pts, scs = [], []
    for i, tok in enumerate(layout):
        item = data[32 * i:32 * i + 32]
        if tok == "P":
            pts.append(item)
        elif tok == "S":
            if int.from_bytes(item, "little") >= R_MOD:            # Fr::from_repr rejects non-canonical encodings
                raise ProofFormatError("invalid field element encoding in proof (item %d)" % i)
            scs.append(item)
        else:
            raise ValueError("layout tokens are 'P' or 'S'")
    return pts, scs


def read_proofs(eng, layout: Sequence[str], proofs: Sequence[bytes]):
    """-> [(points_affine: List[bytes 64], scalars: List[bytes 32])] per proof; one decompression launch for all proofs.
    Raises BadPoint ("invalid point encoding in proof") / ProofFormatError like the reference's reader."""
    split = [split_proof(layout, p) for p in proofs]
    flat = b"".join(b"".join(pts) for pts, _s in split)
    aff = eng.g1_batch_decompress(flat) if flat else b""
    out, k = [], 0
    for pts, scs in split:
        out.append(([aff[64 * (k + j):64 * (k + j) + 64] for j in range(len(pts))], scs))
        k += len(pts)
    return out
