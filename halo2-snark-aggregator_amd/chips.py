"""The reference's plugin boundary for this path — the trait trio ArithCommonChip / ArithFieldChip / ArithEccChip
(halo2-snark-aggregator-api/src/arith/{common,field,ecc}.rs) — as host-side classes over libh2agg.so, with the same
method names, argument meaning and failure behaviour as the Mock chips (mock/arith/{field,ecc}.rs), so code written
against the traits (and the parity tests) runs unchanged on the GPU backend.

Value types, as in the Mock chips:
    AssignedScalar = AssignedNative = Fr      -> 32-byte little-endian canonical integer (`to_repr`)
    AssignedPoint  = C::CurveExt (Jacobian)   -> 96 bytes x || y || z, identity z = 0
    Point          = C (affine)               -> 64 bytes x || y, identity = zeros
`Error` is never constructed by the Mock chips: failure = panic there, an exception here (DivisionByZero for
`invert().unwrap()` mock/arith/field.rs:113, EmptyMultiExp for `acc.unwrap()` mock/arith/ecc.rs:128).

Single-element calls go through the same batch kernels with n = 1: correct, and as slow as a kernel launch — a caller
that cares batches (h2agg_fr_batch_op, h2agg_g1_batch_*) or hands the whole schema over (SchemaBuilder).  No CPU
arithmetic happens here; without the HIP library the constructors fail.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

OP_ADD, OP_SUB, OP_MUL, OP_SQR, OP_INV, OP_DIV = range(6)
_ZERO32 = bytes(32)
_ONE32 = (1).to_bytes(32, "little")
IDENTITY_JAC = bytes(32) + _ONE32 + bytes(32)
GENERATOR_AFF = _ONE32 + (2).to_bytes(32, "little")


class GpuChipCtx:
    """MockChipCtx (mock/arith/field.rs:11-21): `point_list`, `tag`, Display."""

    def __init__(self):
        self.point_list: List[str] = []
        self.tag: str = ""

    def __str__(self):
        return "(total points: %d)" % len(self.point_list)


class GpuFieldChip:
    """ArithFieldChip over Fr (arith/field.rs:6-105; MockFieldChip mock/arith/field.rs:23-146)."""

    def __init__(self, eng):
        self.eng = eng

    # ---- ArithCommonChip (arith/common.rs:3-42)
    def add(self, ctx, a: bytes, b: bytes) -> bytes:                   # mock/arith/field.rs:39-46
        return self.eng.fr_batch_op(OP_ADD, a, b)

    def sub(self, ctx, a: bytes, b: bytes) -> bytes:                   # :48-55
        return self.eng.fr_batch_op(OP_SUB, a, b)

    def assign_zero(self, ctx) -> bytes:
        return _ZERO32

    def assign_one(self, ctx) -> bytes:
        return _ONE32

    def assign_const(self, ctx, c: bytes) -> bytes:
        return bytes(c)

    assign_var = assign_const

    def to_value(self, v: bytes) -> bytes:
        return v

    def normalize(self, ctx, v: bytes) -> bytes:
        return v

    # ---- ArithFieldChip
    def mul(self, ctx, a: bytes, b: bytes) -> bytes:                   # :98-105
        return self.eng.fr_batch_op(OP_MUL, a, b)

    def div(self, ctx, a: bytes, b: bytes) -> bytes:                   # :107-114  a * b.invert().unwrap()
        return self.eng.fr_batch_op(OP_DIV, a, b)

    def square(self, ctx, a: bytes) -> bytes:                          # :116-122
        return self.eng.fr_batch_op(OP_SQR, a)

    def sum_with_coeff_and_constant(self, ctx, a_with_coeff: Sequence[Tuple[bytes, bytes]], b: bytes) -> bytes:
        """acc = b; acc += x * coeff   (mock/arith/field.rs:124-135), one launch"""
        if not a_with_coeff:
            return bytes(b)
        return self.eng.fr_sum_with_coeff_and_constant(b"".join(x for x, _c in a_with_coeff),
                                                       b"".join(c for _x, c in a_with_coeff), b)

    def sum_with_constant(self, ctx, a: Sequence[bytes], b: bytes) -> bytes:          # arith/field.rs:37-48
        return self.sum_with_coeff_and_constant(ctx, [(x, _ONE32) for x in a], b)

    def mul_add_constant(self, ctx, a: bytes, b: bytes, c: bytes) -> bytes:           # mock/arith/field.rs:137-145
        return self.add(ctx, self.mul(ctx, a, b), c)

    def mul_add(self, ctx, a: bytes, b: bytes, c: bytes) -> bytes:                    # arith/field.rs:57-66
        return self.add(ctx, self.mul(ctx, a, b), c)

    def mul_add_accumulate(self, ctx, a: Sequence[bytes], b: bytes) -> bytes:         # arith/field.rs:68-81 (Horner)
        if not a:
            return self.assign_zero(ctx)
        return self.eng.fr_mul_add_accumulate(b"".join(a), b)

    def pow_constant(self, ctx, base: bytes, exponent: int) -> bytes:                 # arith/field.rs:83-104
        assert exponent >= 1
        return self.eng.fr_batch_pow_constant(base, exponent)


class GpuEccChip:
    """ArithEccChip over BN254 G1 (arith/ecc.rs:5-61; MockEccChip mock/arith/ecc.rs:8-130)."""

    def __init__(self, eng):
        self.eng = eng

    def add(self, ctx, a: bytes, b: bytes) -> bytes:                   # mock/arith/ecc.rs:30-37
        return self.eng.g1_batch_add(a, b, subtract=False)

    def sub(self, ctx, a: bytes, b: bytes) -> bytes:                   # :39-46
        return self.eng.g1_batch_add(a, b, subtract=True)

    def assign_zero(self, ctx) -> bytes:                               # :48-50
        return IDENTITY_JAC

    def assign_one(self, ctx) -> bytes:                                # :52-54  generator
        return GENERATOR_AFF + _ONE32

    def assign_const(self, ctx, c_aff: bytes) -> bytes:                # :56-62  to_curve
        return IDENTITY_JAC if c_aff == bytes(64) else bytes(c_aff) + _ONE32

    assign_var = assign_const

    def to_value(self, v_jac: bytes) -> bytes:                         # :64-66  to_affine
        return self.eng.g1_batch_to_affine(v_jac)

    def normalize(self, ctx, v_jac: bytes) -> bytes:                   # :68-74  identity function
        return v_jac

    def scalar_mul(self, ctx, lhs: bytes, rhs_jac: bytes) -> bytes:    # :88-95  rhs * lhs
        return self.eng.g1_batch_scalar_mul(self.eng.g1_batch_to_affine(rhs_jac), lhs)

    def scalar_mul_constant(self, ctx, lhs: bytes, rhs_aff: bytes) -> bytes:          # :97-104
        return self.eng.g1_batch_scalar_mul(rhs_aff, lhs)

    def multi_exp(self, ctx, points: Sequence[bytes], scalars: Sequence[bytes]) -> bytes:
        """mock/arith/ecc.rs:106-129: records the points in ctx.point_list (only their count is observable, through
        Display), then sum_i scalar_i * point_i — here one Pippenger MSM.  Empty input: the reference panics."""
        aff = self.eng.g1_batch_to_affine(b"".join(points)) if points else b""
        ctx.point_list = ["(0x%s, 0x%s)" % (aff[64 * i:64 * i + 32][::-1].hex(), aff[64 * i + 32:64 * i + 64][::-1].hex())
                          if aff[64 * i:64 * i + 64] != bytes(64) else "Infinity" for i in range(len(points))]
        return self.eng.g1_msm(aff, b"".join(scalars))
