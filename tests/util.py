"""Shared helpers for the tests (checker side: uses the oracle)."""
from oracle import bn254 as O, cref

G_BYTES = O.aff_to_bytes(O.G1)


def fr_bytes(xs):
    return b"".join(O.fe_to_bytes(x % O.R) for x in xs)


def rand_frs(rng, n):
    return [rng.fr() for _ in range(n)]


def points_from_scalars(ks):
    """affine bytes of k_i * G via the C oracle"""
    n = len(ks)
    if n == 0:
        return b""
    return cref.g1_batch_to_affine(cref.g1_batch_scalar_mul(G_BYTES * n, fr_bytes(ks), n), n)


def to_jac_bytes(aff_bytes, zs):
    out = b""
    for i, z in enumerate(zs):
        out += O.jac_to_bytes(O.aff_from_bytes(aff_bytes[64 * i:64 * i + 64]), z)
    return out


def norm(eng_or_none, jac):
    """normalise Jacobian bytes with the ORACLE (checker-side normalisation)"""
    return cref.g1_batch_to_affine(jac, len(jac) // 96)
