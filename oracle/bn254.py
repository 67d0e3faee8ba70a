"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Pure-Python big-integer restatement of the arithmetic that sits behind the
reference's "pure calculation context" (MockFieldChip / MockEccChip).  Only
`tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may use it.

PARITY PINNING STATUS: **parity unpinned by reference golden vectors** — the
reference holds no known-answer vectors for this path (every
reference test is an algebraic self-consistency check with time-seeded
randomness) and it cannot be built here (Rust, unvendored git dependencies).
What pins this oracle instead:
  * the two moduli are the literals the reference itself carries
    (halo2-snark-aggregator-solidity/templates/verifier.sol:40-41 `q_mod` = r,
    :143-144 `p_mod` = p, :292 r again in hex);
  * the arithmetic is *exact* prime-field / group arithmetic, so any two
    correct implementations agree bit-for-bit on canonical encodings; this
    file is written from the mathematics (affine chord-and-tangent law with a
    modular inverse), independently of the C restatement (oracle/bn254_ref.c,
    Montgomery + Jacobian) and of the HIP kernels (8x32-bit limbs, XYZZ);
  * the reference's own algebraic identities (halo2-ecc-circuit-lib/src/tests/
    five_native_ecc.rs:60-240) are re-run against it in tests/test_oracle.py;
  * public BN254 known answers (EIP-196 ecAdd/ecMul vectors: 2*G) are checked.

Third-party arithmetic restated: halo2curves 0.2.1 (git tag, commit f75ed26c,
reference Cargo.lock:1569-1571) `bn256::{Fq, Fr, G1, G1Affine}`; not vendored
under /root/reference.  Published algorithm: short-Weierstrass y^2 = x^3 + 3
over Fq, generator (1, 2), prime group order r.

Encodings (shared with include/h2agg.h):
  Fr / Fq     : 32-byte little-endian canonical integer < modulus
  G1 affine   : x || y (64 B); identity = 64 zero bytes  (halo2curves G1Affine identity = (0,0))
  G1 jacobian : x || y || z (96 B); identity has z = 0;  affine = (x / z^2, y / z^3)
"""
from __future__ import annotations

# reference: halo2-snark-aggregator-solidity/templates/verifier.sol:143-144 (p_mod), :40-41 (q_mod)
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
assert P == 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
assert R == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
B = 3
G1 = (1, 2)          # MockEccChip::assign_one -> CurveExt::generator()   mock/arith/ecc.rs:52-54
INF = None           # MockEccChip::assign_zero -> CurveExt::identity()   mock/arith/ecc.rs:48-50


# --------------------------------------------------------------------------- fields
def fr(x: int) -> int:
    return x % R


def fq(x: int) -> int:
    return x % P


def inv(x: int, m: int) -> int:
    """MockFieldChip::div uses `b.invert().unwrap()` (mock/arith/field.rs:107-114): 0 panics."""
    if x % m == 0:
        raise ZeroDivisionError("invert of zero (reference panics: mock/arith/field.rs:113)")
    return pow(x, m - 2, m)


def fr_from_bytes_wide(b: bytes) -> int:
    """`F::from_bytes_wide` as used at mock/transcript_encode.rs:14-21: 512-bit LE integer mod r."""
    assert len(b) == 64
    return int.from_bytes(b, "little") % R


# --------------------------------------------------------------------------- group (affine, exact)
def is_on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def add(a, b):
    """MockEccChip::add = `*a + *b` (mock/arith/ecc.rs:30-37), complete group law."""
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = 3 * x1 * x1 * inv(2 * y1, P) % P
    else:
        lam = (y2 - y1) * inv(x2 - x1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def sub(a, b):
    """MockEccChip::sub = `*a - *b` (mock/arith/ecc.rs:39-46)."""
    return add(a, neg(b))


def double(a):
    return add(a, a)


def scalar_mul(s: int, pt):
    """MockEccChip::scalar_mul = `*rhs * *lhs` (mock/arith/ecc.rs:88-95): halo2curves `G1 * Fr`,
    an MSB-first double-and-add over the 256-bit canonical repr."""
    s %= R
    acc = INF
    for i in range(255, -1, -1):
        acc = double(acc)
        if (s >> i) & 1:
            acc = add(acc, pt)
    return acc


def multi_exp(points, scalars):
    """MockEccChip::multi_exp (mock/arith/ecc.rs:106-129): acc = sum_i scalar_mul(s_i, p_i),
    left to right; empty input panics (`acc.unwrap()`, :128)."""
    if len(points) == 0 or len(scalars) == 0:
        raise ValueError("multi_exp of zero pairs (reference panics: mock/arith/ecc.rs:128)")
    acc = None
    first = True
    for p, s in zip(points, scalars):
        cur = scalar_mul(s, p)
        if first:
            acc, first = cur, False
        else:
            acc = add(acc, cur)
    return acc


# --------------------------------------------------------------------------- encodings
def fe_to_bytes(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def fe_from_bytes(b: bytes) -> int:
    assert len(b) == 32
    return int.from_bytes(b, "little")


def aff_to_bytes(pt) -> bytes:
    if pt is INF:
        return bytes(64)
    return fe_to_bytes(pt[0]) + fe_to_bytes(pt[1])


def aff_from_bytes(b: bytes):
    assert len(b) == 64
    x, y = fe_from_bytes(b[:32]), fe_from_bytes(b[32:])
    if x == 0 and y == 0:
        return INF
    return (x, y)


def compress(pt) -> bytes:
    """G1Affine::to_bytes — the 32-byte point encoding of the proof wire format (written by the prover's transcript,
    read back at systems/halo2/transcript.rs:56-79).  halo2curves 0.2.1 (unvendored; layout recalled from upstream,
    SURVEY.md appendix C): little-endian x, parity of y in bit 7 of byte 31, identity = zeros."""
    if pt is INF:
        return bytes(32)
    b = bytearray(fe_to_bytes(pt[0]))
    b[31] |= (pt[1] & 1) << 7
    return bytes(b)


def decompress(b: bytes):
    """G1Affine::from_bytes.  Returns the point, INF, or raises ValueError ("invalid point encoding in proof")."""
    assert len(b) == 32
    ysign = b[31] >> 7
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x7F]), "little")
    if x >= P:
        raise ValueError("invalid point encoding in proof (x >= p)")
    if x == 0 and not ysign:
        return INF
    rhs = (x * x * x + B) % P
    y = pow(rhs, (P + 1) // 4, P)                # p = 3 mod 4
    if y * y % P != rhs:
        raise ValueError("invalid point encoding in proof (not on the curve)")
    if (y & 1) != ysign:
        y = P - y
    return (x, y)


def jac_to_bytes(pt, z: int = 1) -> bytes:
    """Encode an affine point as Jacobian with the given non-zero z (tests use z != 1 to make sure
    consumers do not assume normalised inputs)."""
    if pt is INF:
        return fe_to_bytes(0) + fe_to_bytes(1) + fe_to_bytes(0)
    z %= P
    assert z != 0
    return fe_to_bytes(pt[0] * z * z % P) + fe_to_bytes(pt[1] * z * z * z % P) + fe_to_bytes(z)


def jac_from_bytes(b: bytes):
    """MockEccChip::to_value = `to_affine` (mock/arith/ecc.rs:64-66)."""
    assert len(b) == 96
    x, y, z = (fe_from_bytes(b[i:i + 32]) for i in (0, 32, 64))
    if z == 0:
        return INF
    zi = inv(z, P)
    zi2 = zi * zi % P
    return (x * zi2 % P, y * zi2 * zi % P)


def debug_fmt(pt) -> str:
    """Stand-in for `format!("{:?}", point)` that MockEccChip::multi_exp stores in ctx.point_list
    (mock/arith/ecc.rs:112-116).  Only the *count* is observable through Display
    (mock/arith/field.rs:17-21); the text format of halo2curves' Debug is not reproduced."""
    if pt is INF:
        return "Infinity"
    return "(0x%064x, 0x%064x)" % pt


# --------------------------------------------------------------------------- deterministic PRNG (shared with tests/bench)
class SplitMix64:
    """splitmix64; the workload generator of BASELINE.md §4 (seed 0x48324147 unless stated)."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def bytes(self, n: int) -> bytes:
        out = b""
        while len(out) < n:
            out += self.next().to_bytes(8, "little")
        return out[:n]

    def fr(self) -> int:
        return fr_from_bytes_wide(self.bytes(64))
