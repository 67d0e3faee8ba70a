"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/bn254_ref.c (liboracle_bn254.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_bn254.so")
_lib = None

OP_ADD, OP_SUB, OP_MUL, OP_SQR, OP_INV = range(5)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bn254_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle_bn254.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _buf(n):
    return (C.c_uint8 * n)()


def field_batch_op(which: int, op: int, a: bytes, b: bytes | None, n: int) -> bytes:
    out = _buf(32 * n)
    rc = lib().oracle_field_batch_op(which, op, a, b, C.c_size_t(n), out)
    if rc:
        raise ZeroDivisionError("oracle_field_batch_op rc=%d" % rc)
    return bytes(out)


def fr_mul_add_accumulate(v: bytes, n: int, b: bytes) -> bytes:
    out = _buf(32)
    lib().oracle_fr_mul_add_accumulate(v, C.c_size_t(n), b, out)
    return bytes(out)


def g1_batch_add(a: bytes, b: bytes, n: int, subtract: bool = False) -> bytes:
    out = _buf(96 * n)
    lib().oracle_g1_batch_add(a, b, C.c_size_t(n), int(subtract), out)
    return bytes(out)


def g1_batch_scalar_mul(bases: bytes, scalars: bytes, n: int) -> bytes:
    out = _buf(96 * n)
    lib().oracle_g1_batch_scalar_mul(bases, scalars, C.c_size_t(n), out)
    return bytes(out)


def g1_batch_to_affine(jac: bytes, n: int) -> bytes:
    out = _buf(64 * n)
    lib().oracle_g1_batch_to_affine(jac, C.c_size_t(n), out)
    return bytes(out)


def multi_exp_naive(bases: bytes, scalars: bytes, n: int) -> bytes:
    out = _buf(64)
    rc = lib().oracle_multi_exp_naive(bases, scalars, C.c_size_t(n), out)
    if rc:
        raise ValueError("multi_exp of zero pairs (reference panics)")
    return bytes(out)


def eval_flat(pts: bytes, scalars: bytes, has_scalar: bytes, n: int) -> bytes:
    out = _buf(64)
    rc = lib().oracle_eval_flat(pts, scalars, has_scalar, C.c_size_t(n), out)
    if rc:
        raise ValueError("eval_flat without any scalar-carrying point (reference panics)")
    return bytes(out)


def msm_pippenger(bases: bytes, scalars: bytes, n: int, c: int = 0, nthreads: int = 1) -> bytes:
    if c == 0:
        c = max(2, min(16, n.bit_length() - 3))
    out = _buf(64)
    rc = lib().oracle_msm_pippenger(bases, scalars, C.c_size_t(n), c, nthreads, out)
    assert rc == 0
    return bytes(out)


def msm_pippenger2(bases: bytes, scalars: bytes, n: int, c: int = 0, nthreads: int = 1, jobs_per_thread: int = 4) -> bytes:
    """the competent CPU Pippenger (signed digits, XYZZ buckets with mixed additions, balanced (window, range) jobs)"""
    if c == 0:
        c = max(2, min(16, n.bit_length() - 4))
    out = _buf(64)
    rc = lib().oracle_msm_pippenger2(bases, scalars, C.c_size_t(n), c, nthreads, jobs_per_thread, out)
    assert rc == 0
    return bytes(out)


def constants(which: int):
    mod, r1, r2 = _buf(32), _buf(32), _buf(32)
    inv = C.c_uint64()
    lib().oracle_constants(which, mod, r1, r2, C.byref(inv))
    return (int.from_bytes(bytes(mod), "little"), int.from_bytes(bytes(r1), "little"),
            int.from_bytes(bytes(r2), "little"), inv.value)
