"""halo2-snark-aggregator_amd — MI355X (gfx950) backend for the pure-calculation hot path of
scroll-tech/halo2-snark-aggregator (BN254 G1 multi_exp + multi-open evaluation).

This module is a thin ctypes binding of the C ABI in include/h2agg.h (libh2agg.so, built from
csrc/*.hip by build_ext.py).  All arithmetic happens in HIP kernels; there is NO CPU fallback:
`load_library()` raises if the shared object is missing and `H2Agg()` raises if no HIP device is
usable.  The directory name contains a hyphen (it mirrors the reference's crate naming), so import it
with `importlib` — see tests/conftest.py / __graft_entry__.py (`load_package()`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libh2agg.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "h2agg.h")

OK, ERR_INVALID, ERR_DIV_ZERO, ERR_EMPTY, ERR_HIP, ERR_NONCANONICAL, ERR_NOMEM, ERR_BAD_POINT, ERR_PEER = range(9)
OP_ADD, OP_SUB, OP_MUL, OP_SQR, OP_INV, OP_DIV = range(6)

IDENTITY_JAC = (0).to_bytes(32, "little") + (1).to_bytes(32, "little") + (0).to_bytes(32, "little")


class H2AggError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("h2agg error %d: %s" % (code, msg))
        self.code = code


class EmptyMultiExp(H2AggError):
    """multi_exp of zero pairs — the reference panics (`acc.unwrap()`, mock/arith/ecc.rs:128)."""


class BadPoint(H2AggError, ValueError):
    """"invalid point encoding in proof" (systems/halo2/transcript.rs:65-70)."""


class DivisionByZero(H2AggError, ZeroDivisionError):
    """inversion of zero — the reference panics (`invert().unwrap()`, mock/arith/field.rs:113)."""


_lib = None


def load_library():
    """dlopen libh2agg.so and declare every prototype.  Fails loudly if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libh2agg.so is not built (%s). Run `python __graft_entry__.py build` — the HIP extension is "
            "required, there is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own copy of the HIP runtime.  Whichever copy is loaded first serves the whole process: with
    # libh2agg.so first (the system's /opt/rocm runtime), a later `import torch` finds "No HIP GPUs are available".  The
    # harness uses torch for device memory and streams anyway, so let it load its runtime before the library binds to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    u8p, vp, sz, i32, u64 = C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    ctxp = C.c_void_p
    protos = {
        "h2agg_create": (i32, [i32, C.POINTER(ctxp)]),
        "h2agg_destroy": (None, [ctxp]),
        "h2agg_last_error": (C.c_char_p, [ctxp]),
        "h2agg_set_stream": (i32, [ctxp, vp]),
        "h2agg_synchronize": (i32, [ctxp]),
        "h2agg_describe": (C.c_char_p, [ctxp]),
        "h2agg_fr_batch_op": (i32, [ctxp, i32, u8p, u8p, sz, vp]),
        "h2agg_fr_batch_pow_constant": (i32, [ctxp, u8p, sz, u64, vp]),
        "h2agg_fr_tape_eval": (i32, [ctxp, u8p, sz, vp, sz, vp, sz, vp]),
        "h2agg_fr_mul_add_accumulate": (i32, [ctxp, u8p, sz, u8p, vp]),
        "h2agg_fr_sum_with_coeff_and_constant": (i32, [ctxp, u8p, u8p, sz, u8p, vp]),
        "h2agg_g1_batch_add": (i32, [ctxp, u8p, u8p, sz, i32, vp]),
        "h2agg_g1_batch_scalar_mul": (i32, [ctxp, u8p, u8p, sz, vp]),
        "h2agg_g1_batch_to_affine": (i32, [ctxp, u8p, sz, vp]),
        "h2agg_g1_sum": (i32, [ctxp, u8p, sz, vp]),
        "h2agg_g1_batch_decompress": (i32, [ctxp, u8p, sz, vp, vp]),
        "h2agg_g1_batch_compress": (i32, [ctxp, u8p, sz, vp]),
        "h2agg_g1_msm": (i32, [ctxp, vp, vp, sz, vp]),
        "h2agg_g1_msm_jac": (i32, [ctxp, vp, vp, sz, vp]),
        "h2agg_host_alloc": (i32, [ctxp, sz, C.POINTER(vp)]),
        "h2agg_host_free": (i32, [ctxp, vp]),
        "h2agg_eval_flat": (i32, [ctxp, u8p, u8p, u8p, sz, vp]),
        "h2agg_bases_upload": (i32, [ctxp, u8p, sz, C.POINTER(u64)]),
        "h2agg_bases_generate": (i32, [ctxp, vp, sz, C.POINTER(u64)]),
        "h2agg_bases_precompute": (i32, [ctxp, u64, i32]),
        "h2agg_bases_download": (i32, [ctxp, u64, sz, sz, vp]),
        "h2agg_bases_free": (i32, [ctxp, u64]),
        "h2agg_g1_msm_preloaded": (i32, [ctxp, u64, u8p, sz, vp]),
        "h2agg_instance_commitment": (i32, [ctxp, u64, u8p, sz, sz, vp]),
        "h2agg_g1_msm_device": (i32, [ctxp, u64, vp, sz, vp]),
        "h2agg_g1_msm_device_async": (i32, [ctxp, u64, vp, sz, vp]),
        "h2agg_schema_create": (i32, [ctxp, C.POINTER(C.c_void_p)]),
        "h2agg_schema_destroy": (None, [C.c_void_p]),
        "h2agg_schema_node_commitment": (i32, [C.c_void_p, C.c_char_p, u8p, C.POINTER(C.c_uint32)]),
        "h2agg_schema_node_eval": (i32, [C.c_void_p, u8p, C.POINTER(C.c_uint32)]),
        "h2agg_schema_node_scalar": (i32, [C.c_void_p, u8p, C.POINTER(C.c_uint32)]),
        "h2agg_schema_node_add": (i32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
        "h2agg_schema_node_mul": (i32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
        "h2agg_schema_estimate": (i32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t)]),
        "h2agg_schema_evaluation_queries": (i32, [C.c_void_p, sz, C.POINTER(C.c_char_p), u8p, u8p,
                                                  C.POINTER(C.c_uint32)]),
        "h2agg_schema_batch_multi_open": (i32, [C.c_void_p, C.c_char_p, sz, C.POINTER(C.c_int32), u8p,
                                                C.POINTER(C.c_uint32), sz, u8p, u8p, u8p,
                                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "h2agg_schema_eval": (i32, [C.c_void_p, C.c_uint32, vp, C.POINTER(i32), vp]),
        "h2agg_evaluate_multiopen_proof": (i32, [C.c_void_p, C.c_uint32, C.c_uint32, vp, vp]),
        "h2agg_evaluate_multiopen_prepare": (i32, [C.c_void_p, C.c_uint32, C.c_uint32]),
        "h2agg_schema_name_count": (C.c_size_t, [C.c_void_p]),
        "h2agg_schema_name": (C.c_char_p, [C.c_void_p, C.c_size_t]),
        "h2agg_g1_msm_device_batch_async": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
        "h2agg_schema_query_set_commitment": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
        "h2agg_g1_batch_to_affine_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]),
        "h2agg_schema_names_joined": (C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
        "h2agg_schema_point_list_len": (C.c_size_t, [C.c_void_p]),
        "h2agg_poseidon_squeeze_batch": (i32, [ctxp, u8p, sz, sz, C.POINTER(C.c_uint32), sz, vp]),
        "h2agg_transcript_read_batch": (i32, [ctxp, u8p, sz, sz, C.c_char_p, sz, u8p, sz, u8p, sz, vp, vp]),
        "h2agg_vk_create": (i32, [ctxp, u8p, sz, C.POINTER(C.c_void_p)]),
        "h2agg_vk_destroy": (None, [C.c_void_p]),
        "h2agg_verify_aggregation": (i32, [ctxp, vp, sz, u8p, u8p, vp, vp, vp, C.POINTER(i32)]),   # see verifier.py
        "h2agg_verify_aggregation_ex": (i32, [ctxp, vp, sz, u8p, u8p, vp, vp, vp, C.POINTER(i32), vp, sz]),
        "h2agg_verify_aggregation_sharded": (i32, [ctxp, vp, sz, vp, u8p, u8p, vp, vp, vp, C.POINTER(i32), vp, sz]),   # see verifier.py
        "h2agg_debug_configure": (i32, [ctxp, C.c_char_p, i32]),
        "h2agg_verify_plan_stats": (i32, [ctxp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "h2agg_last_phases": (C.c_char_p, [ctxp]),
        "h2agg_transcript_configure": (i32, [ctxp, i32]),
        "h2agg_poseidon_squeeze_batch_host": (i32, [u8p, sz, sz, C.POINTER(C.c_uint32), sz, vp, i32]),
        "h2agg_host_threads": (i32, []),
        "h2agg_host_sponge_kind": (i32, []),
        "h2agg_comm_unique_id": (i32, [vp]),
        "h2agg_comm_init_rank": (i32, [ctxp, u8p, i32, i32]),
        "h2agg_comm_create": (i32, [C.POINTER(i32), i32, C.POINTER(ctxp)]),
        "h2agg_comm_size": (i32, [ctxp]),
        "h2agg_comm_rank": (i32, [ctxp]),
        "h2agg_comm_last_error": (C.c_char_p, []),
        "h2agg_allgather_add_points": (i32, [C.POINTER(ctxp), i32, u8p, sz, vp]),
        "h2agg_pairing_check": (i32, [ctxp, u8p, u8p, sz, C.POINTER(i32)]),
        "h2agg_g2_batch_decompress": (i32, [ctxp, u8p, sz, vp]),
        "h2agg_pairing_product": (i32, [ctxp, u8p, u8p, sz, vp]),
        "h2agg_final_pair_check": (i32, [ctxp, u8p, u8p, u8p, u8p, C.POINTER(i32)]),
        "h2agg_msm_configure": (i32, [ctxp, i32, i32, i32]),
        "h2agg_msm_configure_glv": (i32, [ctxp, i32]),
        "h2agg_msm_configure_lanes_per_bucket": (i32, [ctxp, i32]),
        "h2agg_msm_configure_sort": (i32, [ctxp, i32, i32]),
        "h2agg_msm_set_tail_overlap": (i32, [ctxp, i32]),
        "h2agg_profile_enable": (i32, [ctxp, i32]),
        "h2agg_profile_reset": (i32, [ctxp]),
        "h2agg_profile_stage_count": (i32, [ctxp]),
        "h2agg_profile_stage_name": (C.c_char_p, [ctxp, i32]),
        "h2agg_profile_stage_get": (i32, [ctxp, i32, C.POINTER(C.c_double), C.POINTER(u64)]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib._h2agg_protos = tuple(protos)
    _lib = lib
    return lib


def exported_symbols() -> Sequence[str]:
    return load_library()._h2agg_protos


def _need(buf, nbytes: int, what: str):
    """the C side trusts the lengths it is given: a short Python buffer would be read past its end"""
    if buf is None or len(buf) != nbytes:
        raise ValueError("%s must be exactly %d bytes (got %s)" % (what, nbytes, "None" if buf is None else len(buf)))


class H2Agg:
    """One context = one GPU, single-thread-affine (mirrors `MockEccChip::default()` +
    `MockFieldChip::default()` + `MockChipCtx::default()`, verify_circuit.rs:115-118)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.h2agg_create(int(device), C.byref(self._ctx))
        if rc != OK:
            self._ctx = None
            raise H2AggError(rc, "h2agg_create(device=%d) failed: a HIP device is required (no CPU mode)" % device)
        self.device = device

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_ctx", None):
            for ref in list(getattr(self, "_builders", ())):   # schemas hold device buffers of this context
                b = ref()
                if b is not None:
                    b.close()
            self._lib.h2agg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _register_builder(self, b):
        import weakref
        if not hasattr(self, "_builders"):
            self._builders = []
        self._builders = [r for r in self._builders if r() is not None]
        self._builders.append(weakref.ref(b))

    def _check(self, rc: int):
        if rc == OK:
            return
        msg = (self._lib.h2agg_last_error(self._ctx) or b"").decode()
        if rc == ERR_EMPTY:
            raise EmptyMultiExp(rc, msg)
        if rc == ERR_DIV_ZERO:
            raise DivisionByZero(rc, msg)
        if rc == ERR_BAD_POINT:
            raise BadPoint(rc, msg)
        raise H2AggError(rc, msg)

    def describe(self) -> str:
        return self._lib.h2agg_describe(self._ctx).decode()

    def set_stream(self, stream_ptr: Optional[int]):
        """a hipStream_t handle for every launch of this context; None or 0 = the context's OWN non-blocking stream.  torch's
        default stream has the handle 0: `set_stream(torch.cuda.current_stream().cuda_stream)` therefore does NOT put the
        library on torch's stream — make a real one (`torch.cuda.Stream()`, `torch.cuda.set_stream`) or synchronise before
        handing device buffers over (include/h2agg.h)."""
        self._check(self._lib.h2agg_set_stream(self._ctx, stream_ptr))

    def synchronize(self):
        self._check(self._lib.h2agg_synchronize(self._ctx))

    # ------------------------------------------------------------------ Fr
    def fr_batch_op(self, op: int, a: bytes, b: Optional[bytes] = None) -> bytes:
        n = len(a) // 32
        _need(a, 32 * n, "a")
        if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV):
            _need(b, 32 * n, "b")
        out = C.create_string_buffer(32 * n) if n else C.create_string_buffer(1)
        self._check(self._lib.h2agg_fr_batch_op(self._ctx, op, a, b, n, out))
        return out.raw[:32 * n]

    def fr_batch_pow_constant(self, a: bytes, exponent: int) -> bytes:
        n = len(a) // 32
        _need(a, 32 * n, "a")
        out = C.create_string_buffer(max(32 * n, 1))
        self._check(self._lib.h2agg_fr_batch_pow_constant(self._ctx, a, n, exponent, out))
        return out.raw[:32 * n]

    def fr_tape_eval(self, consts: bytes, ops: Sequence, out_regs: Sequence[int]) -> bytes:
        """ops: (opcode, a, b) triples (0 = mul, 1 = add, 2 = sub); registers: inputs first, then one per op"""
        nconst, nops, nout = len(consts) // 32, len(ops), len(out_regs)
        _need(consts, 32 * nconst, "consts")
        flat = (C.c_uint32 * max(3 * nops, 1))(*[x for op in ops for x in op])
        regs = (C.c_uint32 * max(nout, 1))(*out_regs)
        out = C.create_string_buffer(max(32 * nout, 1))
        self._check(self._lib.h2agg_fr_tape_eval(self._ctx, consts, nconst, flat, nops, regs, nout, out))
        return out.raw[:32 * nout]

    def fr_mul_add_accumulate(self, v: bytes, b: bytes) -> bytes:
        _need(v, 32 * (len(v) // 32), "v")
        _need(b, 32, "b")
        out = C.create_string_buffer(32)
        self._check(self._lib.h2agg_fr_mul_add_accumulate(self._ctx, v, len(v) // 32, b, out))
        return out.raw

    def fr_sum_with_coeff_and_constant(self, x: bytes, coeff: bytes, b: bytes) -> bytes:
        _need(x, 32 * (len(x) // 32), "x")
        _need(coeff, len(x), "coeff")
        _need(b, 32, "b")
        out = C.create_string_buffer(32)
        self._check(self._lib.h2agg_fr_sum_with_coeff_and_constant(self._ctx, x, coeff, len(x) // 32, b, out))
        return out.raw

    # ------------------------------------------------------------------ G1 batch
    def g1_batch_add(self, a_jac: bytes, b_jac: bytes, subtract: bool = False) -> bytes:
        n = len(a_jac) // 96
        _need(a_jac, 96 * n, "a_jac")
        _need(b_jac, 96 * n, "b_jac")
        out = C.create_string_buffer(max(96 * n, 1))
        self._check(self._lib.h2agg_g1_batch_add(self._ctx, a_jac, b_jac, n, int(subtract), out))
        return out.raw[:96 * n]

    def g1_batch_scalar_mul(self, bases_aff: bytes, scalars: bytes) -> bytes:
        n = len(scalars) // 32
        _need(scalars, 32 * n, "scalars")
        _need(bases_aff, 64 * n, "bases_aff")
        out = C.create_string_buffer(max(96 * n, 1))
        self._check(self._lib.h2agg_g1_batch_scalar_mul(self._ctx, bases_aff, scalars, n, out))
        return out.raw[:96 * n]

    def g1_batch_to_affine(self, jac: bytes) -> bytes:
        n = len(jac) // 96
        _need(jac, 96 * n, "jac")
        out = C.create_string_buffer(max(64 * n, 1))
        self._check(self._lib.h2agg_g1_batch_to_affine(self._ctx, jac, n, out))
        return out.raw[:64 * n]

    def g1_batch_to_affine_device(self, d_jac_ptr: int, n: int) -> bytes:
        out = C.create_string_buffer(64 * n)
        self._check(self._lib.h2agg_g1_batch_to_affine_device(self._ctx, C.c_void_p(d_jac_ptr), n, out))
        return out.raw

    def g1_batch_decompress(self, data: bytes, with_ok: bool = False):
        """proof wire format -> canonical affine (transcript.rs:63-70); raises BadPoint unless with_ok"""
        n = len(data) // 32
        _need(data, 32 * n, "data")
        out, ok = C.create_string_buffer(max(64 * n, 1)), C.create_string_buffer(max(n, 1))
        rc = self._lib.h2agg_g1_batch_decompress(self._ctx, data, n, out, ok)
        if with_ok and rc in (OK, ERR_BAD_POINT):
            return out.raw[:64 * n], ok.raw[:n]
        self._check(rc)
        return out.raw[:64 * n]

    def g1_batch_compress(self, aff: bytes) -> bytes:
        n = len(aff) // 64
        _need(aff, 64 * n, "aff")
        out = C.create_string_buffer(max(32 * n, 1))
        self._check(self._lib.h2agg_g1_batch_compress(self._ctx, aff, n, out))
        return out.raw[:32 * n]

    def g1_sum(self, jac: bytes) -> bytes:
        _need(jac, 96 * (len(jac) // 96), "jac")
        out = C.create_string_buffer(96)
        self._check(self._lib.h2agg_g1_sum(self._ctx, jac, len(jac) // 96, out))
        return out.raw

    # ------------------------------------------------------------------ MSM
    def g1_msm(self, bases_aff, scalars, n: Optional[int] = None) -> bytes:
        """bases_aff / scalars: bytes, or addresses of host buffers (e.g. from host_alloc) together with n"""
        out = C.create_string_buffer(96)
        if n is None:
            n = len(scalars) // 32
        if isinstance(scalars, (bytes, bytearray)):
            _need(scalars, 32 * n, "scalars")
        if isinstance(bases_aff, (bytes, bytearray)):
            _need(bases_aff, 64 * n, "bases_aff")
        pb = C.cast(bases_aff, C.c_void_p) if isinstance(bases_aff, (bytes, bytearray)) else C.c_void_p(bases_aff)
        ps = C.cast(scalars, C.c_void_p) if isinstance(scalars, (bytes, bytearray)) else C.c_void_p(scalars)
        self._check(self._lib.h2agg_g1_msm(self._ctx, pb, ps, n, out))
        return out.raw

    def g1_msm_jac(self, points_jac: bytes, scalars: bytes) -> bytes:
        """multi_exp over projective points (x || y || z, 96 B each): normalised on the device"""
        n = len(scalars) // 32
        _need(scalars, 32 * n, "scalars")
        _need(points_jac, 96 * n, "points_jac")
        out = C.create_string_buffer(96)
        self._check(self._lib.h2agg_g1_msm_jac(self._ctx, C.cast(points_jac, C.c_void_p), C.cast(scalars, C.c_void_p), n, out))
        return out.raw

    def host_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.h2agg_host_alloc(self._ctx, nbytes, C.byref(p)))
        return p.value

    def host_free(self, ptr: int):
        self._check(self._lib.h2agg_host_free(self._ctx, C.c_void_p(ptr)))

    def eval_flat(self, pts_aff: bytes, scalars: bytes, has_scalar: bytes) -> bytes:
        n = len(has_scalar)
        _need(pts_aff, 64 * n, "pts_aff")
        _need(scalars, 32 * n, "scalars")
        out = C.create_string_buffer(96)
        self._check(self._lib.h2agg_eval_flat(self._ctx, pts_aff, scalars, has_scalar, len(has_scalar), out))
        return out.raw

    def bases_upload(self, bases_aff: bytes) -> int:
        _need(bases_aff, 64 * (len(bases_aff) // 64), "bases_aff")
        h = C.c_uint64()
        self._check(self._lib.h2agg_bases_upload(self._ctx, bases_aff, len(bases_aff) // 64, C.byref(h)))
        return h.value

    def bases_generate(self, d_k_scalars_ptr: int, n: int) -> int:
        h = C.c_uint64()
        self._check(self._lib.h2agg_bases_generate(self._ctx, d_k_scalars_ptr, n, C.byref(h)))
        return h.value

    def bases_precompute(self, handle: int, window_bits: int = 0):
        """fixed-base levels for a resident table (SRS-style bases): later MSMs over it use one bucket set"""
        self._check(self._lib.h2agg_bases_precompute(self._ctx, handle, window_bits))

    def bases_download(self, handle: int, first: int, n: int) -> bytes:
        out = C.create_string_buffer(max(64 * n, 1))
        self._check(self._lib.h2agg_bases_download(self._ctx, handle, first, n, out))
        return out.raw[:64 * n]

    def bases_free(self, handle: int):
        self._check(self._lib.h2agg_bases_free(self._ctx, handle))

    def g1_msm_preloaded(self, handle: int, scalars: bytes) -> bytes:
        _need(scalars, 32 * (len(scalars) // 32), "scalars")
        out = C.create_string_buffer(96)
        self._check(self._lib.h2agg_g1_msm_preloaded(self._ctx, handle, scalars, len(scalars) // 32, out))
        return out.raw

    def instance_commitment(self, g_lagrange_handle: int, instance: bytes, max_len: int) -> bytes:
        _need(instance, 32 * (len(instance) // 32), "instance")
        out = C.create_string_buffer(96)
        self._check(self._lib.h2agg_instance_commitment(self._ctx, g_lagrange_handle, instance, len(instance) // 32,
                                                        max_len, out))
        return out.raw

    def g1_msm_device(self, handle: int, d_scalars_ptr: int, n: int) -> bytes:
        out = C.create_string_buffer(96)
        self._check(self._lib.h2agg_g1_msm_device(self._ctx, handle, d_scalars_ptr, n, out))
        return out.raw

    def g1_msm_device_async(self, handle: int, d_scalars_ptr: int, n: int, d_out_ptr: int):
        self._check(self._lib.h2agg_g1_msm_device_async(self._ctx, handle, d_scalars_ptr, n, d_out_ptr))

    # ------------------------------------------------------------------ transcript side
    def poseidon_squeeze_batch(self, elems: bytes, nproofs: int, upto: Sequence[int]) -> bytes:
        """nproofs sponges (T=9, RATE=8, R_F=8, R_P=63): elems [nproofs][nelem] canonical Fr -> [nproofs][len(upto)] challenges"""
        nelem = (len(elems) // 32) // max(nproofs, 1)
        _need(elems, 32 * nelem * nproofs, "elems")
        nsq = len(upto)
        arr = (C.c_uint32 * max(nsq, 1))(*upto)
        out = C.create_string_buffer(max(32 * nproofs * nsq, 1))
        self._check(self._lib.h2agg_poseidon_squeeze_batch(self._ctx, elems, nproofs, nelem, arr, nsq, out))
        return out.raw[:32 * nproofs * nsq]

    def verify_plan_stats(self):
        """(hits, misses, plans kept) of the context's recorded-aggregation cache (h2agg_verify_plan_stats)"""
        h, m, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.h2agg_verify_plan_stats(self._ctx, C.byref(h), C.byref(m), C.byref(k)))
        return h.value, m.value, k.value

    def last_phases(self) -> str:
        """wall-clock split of the last verify_aggregation call (after debug_configure("phases", 1)): h2agg_last_phases"""
        return (self._lib.h2agg_last_phases(self._ctx) or b"").decode()

    def transcript_configure(self, backend: str = "auto"):
        """which backend runs the Poseidon sponges of this context: "auto" (by batch size), "device", "host" (worker threads)"""
        self._check(self._lib.h2agg_transcript_configure(self._ctx, {"auto": 0, "device": 1, "host": 2}[backend]))

    def transcript_read_batch(self, proofs: Sequence[bytes], script: str, consts: bytes = b"", ext_points_aff: bytes = b""):
        """PoseidonTranscriptRead over same-layout proofs -> (points [proof] bytes, challenges [proof] bytes)"""
        nproofs = len(proofs)
        plen = 32 * (script.count("P") + script.count("S"))
        for p in proofs:
            _need(p, plen, "proof")
        npts, nsq, nx = script.count("P"), script.count("Q"), script.count("X")
        _need(consts, 32 * script.count("C"), "consts")
        _need(ext_points_aff, 64 * nx * nproofs, "ext_points_aff")
        pts = C.create_string_buffer(max(64 * npts * nproofs, 1))
        ch = C.create_string_buffer(max(32 * nsq * nproofs, 1))
        sb = script.encode()
        self._check(self._lib.h2agg_transcript_read_batch(self._ctx, b"".join(proofs), plen, nproofs, sb, len(sb), consts,
                                                          len(consts) // 32, ext_points_aff, nx, pts, ch))
        return ([pts.raw[64 * npts * i:64 * npts * (i + 1)] for i in range(nproofs)],
                [ch.raw[32 * nsq * i:32 * nsq * (i + 1)] for i in range(nproofs)])

    # ------------------------------------------------------------------ multi-GPU exchange (RCCL inside the C ABI)
    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the 128-byte RCCL id every rank passes to comm_init_rank"""
        out = C.create_string_buffer(128)
        rc = load_library().h2agg_comm_unique_id(out)
        if rc != OK:
            why = load_library().h2agg_comm_last_error() or b""
            raise H2AggError(rc, "h2agg_comm_unique_id failed: " + (why.decode(errors="replace") or "RCCL error"))
        return out.raw

    def comm_init_rank(self, unique_id: bytes, rank: int, nranks: int):
        _need(unique_id, 128, "unique_id")
        self._check(self._lib.h2agg_comm_init_rank(self._ctx, unique_id, rank, nranks))

    def comm_size(self) -> int:
        """ranks of this context's communicator (0 = none)"""
        return self._lib.h2agg_comm_size(self._ctx)

    def comm_rank(self) -> int:
        """this context's rank in its communicator (-1 = none)"""
        return self._lib.h2agg_comm_rank(self._ctx)

    def allgather_add_points(self, partial_jac: bytes) -> bytes:
        """this rank's partial accumulators (96 B each) -> affine sums over all ranks (64 B each), same on every rank:
        the one collective of the sharded aggregation (all-gather over RCCL + local EC adds)"""
        npts = len(partial_jac) // 96
        _need(partial_jac, 96 * npts, "partial_jac")
        out = C.create_string_buffer(max(64 * npts, 1))
        arr = (C.c_void_p * 1)(self._ctx)
        self._check(self._lib.h2agg_allgather_add_points(arr, 1, partial_jac, npts, out))
        return out.raw[:64 * npts]

    # ------------------------------------------------------------------ pairing (host)
    def pairing_check(self, g1_aff: bytes, g2_aff: bytes) -> bool:
        """prod e(g1_i, g2_i) == 1  (verify.rs:733-739 / EIP-197); G2 = x.c0 || x.c1 || y.c0 || y.c1"""
        n = len(g1_aff) // 64
        _need(g1_aff, 64 * n, "g1_aff")
        _need(g2_aff, 128 * n, "g2_aff")
        ok = C.c_int()
        self._check(self._lib.h2agg_pairing_check(self._ctx, g1_aff, g2_aff, n, C.byref(ok)))
        return bool(ok.value)

    def pairing_product(self, g1_aff: bytes, g2_aff: bytes) -> bytes:
        n = len(g1_aff) // 64
        _need(g1_aff, 64 * n, "g1_aff")
        _need(g2_aff, 128 * n, "g2_aff")
        out = C.create_string_buffer(384)
        self._check(self._lib.h2agg_pairing_product(self._ctx, g1_aff, g2_aff, n, out))
        return out.raw

    def g2_batch_decompress(self, data: bytes) -> bytes:
        """64-byte compressed G2 points (ParamsKZG::write's g2 / s_g2) -> 128-byte affine"""
        n = len(data) // 64
        _need(data, 64 * n, "data")
        out = C.create_string_buffer(max(128 * n, 1))
        self._check(self._lib.h2agg_g2_batch_decompress(self._ctx, data, n, out))
        return out.raw[:128 * n]

    def final_pair_check(self, left_aff: bytes, right_aff: bytes, s_g2: bytes, g2: bytes) -> bool:
        """e(left, [s]_2) * e(right, -[1]_2) == 1"""
        _need(left_aff, 64, "left_aff")
        _need(right_aff, 64, "right_aff")
        _need(s_g2, 128, "s_g2")
        _need(g2, 128, "g2")
        ok = C.c_int()
        self._check(self._lib.h2agg_final_pair_check(self._ctx, left_aff, right_aff, s_g2, g2, C.byref(ok)))
        return bool(ok.value)

    # ------------------------------------------------------------------ tuning / measurement
    def g1_msm_device_batch_async(self, handle: int, d_scalars_ptr: int, n: int, batch: int, d_out_ptr: int):
        self._check(self._lib.h2agg_g1_msm_device_batch_async(self._ctx, handle, C.c_void_p(d_scalars_ptr), n, batch,
                                                              C.c_void_p(d_out_ptr)))

    def msm_configure(self, window_bits: int = 0, reduce_segment: int = 0, big_bucket_threshold: int = 0):
        self._check(self._lib.h2agg_msm_configure(self._ctx, window_bits, reduce_segment, big_bucket_threshold))

    def msm_configure_glv(self, mode: int = 0):
        self._check(self._lib.h2agg_msm_configure_glv(self._ctx, mode))

    def msm_configure_lanes_per_bucket(self, lanes: int = 0):
        self._check(self._lib.h2agg_msm_configure_lanes_per_bucket(self._ctx, lanes))

    def msm_configure_sort(self, sub_bits: int = 0, tile: int = 0):
        self._check(self._lib.h2agg_msm_configure_sort(self._ctx, sub_bits, tile))

    def debug_configure(self, key: str, value: int):
        """h2agg_debug_configure: per-context test hooks (include/h2agg.h "Environment")"""
        self._check(self._lib.h2agg_debug_configure(self._ctx, key.encode(), int(value)))

    def msm_set_tail_overlap(self, level: int = 2):
        """0 = off, 1 = only the Horner kernel on a tail stream, 2 = bucket reduction + window sums + Horner"""
        self._check(self._lib.h2agg_msm_set_tail_overlap(self._ctx, int(level)))

    def profile_enable(self, on=True, only_stage: Optional[int] = None):
        """only_stage: index into profile_stages() order -> bracket just that stage (cheaper)."""
        mode = (2 + int(only_stage)) if (on and only_stage is not None) else int(bool(on))
        self._check(self._lib.h2agg_profile_enable(self._ctx, mode))

    def profile_reset(self):
        self._check(self._lib.h2agg_profile_reset(self._ctx))

    def profile_stages(self):
        """-> {stage name: (total_ms, launches)}"""
        out = {}
        for i in range(self._lib.h2agg_profile_stage_count(self._ctx)):
            ms, cnt = C.c_double(), C.c_uint64()
            self._check(self._lib.h2agg_profile_stage_get(self._ctx, i, C.byref(ms), C.byref(cnt)))
            out[self._lib.h2agg_profile_stage_name(self._ctx, i).decode()] = (ms.value, cnt.value)
        return out


def host_sponge_kind() -> str:
    """arithmetic the host sponge runs on here: "ifma" (AVX-512 IFMA) or "scalar" (portable 4 x 64-bit)"""
    return "ifma" if load_library().h2agg_host_sponge_kind() == 1 else "scalar"


def poseidon_squeeze_batch_host(elems: bytes, nproofs: int, upto: Sequence[int], max_threads: int = 0,
                                kernel: Optional[str] = None) -> bytes:
    """the host backend of H2Agg.poseidon_squeeze_batch on its own (h2agg_poseidon_squeeze_batch_host: no context, no device);
    kernel: None = the process-wide choice, "scalar" / "ifma" = force one for this call (ifma only where the CPU has it)"""
    lib = load_library()
    max_threads = (max_threads & 0xffff) | {None: 0, "scalar": 0x10000, "ifma": 0x20000}[kernel]
    nelem = (len(elems) // 32) // max(nproofs, 1)
    _need(elems, 32 * nelem * nproofs, "elems")
    nsq = len(upto)
    arr = (C.c_uint32 * max(nsq, 1))(*upto)
    out = C.create_string_buffer(32 * max(nproofs * nsq, 1))
    rc = lib.h2agg_poseidon_squeeze_batch_host(elems, nproofs, nelem, arr, nsq, out, max_threads)
    if rc != OK:
        raise H2AggError(rc, "h2agg_poseidon_squeeze_batch_host: " + ("element >= r" if rc == ERR_NONCANONICAL else "invalid arguments"))
    return out.raw[:32 * nproofs * nsq]


def host_threads() -> int:
    """worker threads the library may use for per-proof host work (h2agg_host_threads)"""
    return load_library().h2agg_host_threads()


class H2AggGroup:
    """One process driving several GPUs: h2agg_comm_create = a context per device + ncclCommInitAll."""

    def __init__(self, devices: Sequence[int]):
        lib = load_library()
        n = len(devices)
        devs = (C.c_int * n)(*devices)
        ctxs = (C.c_void_p * n)()
        rc = lib.h2agg_comm_create(devs, n, ctxs)
        if rc != OK:
            msg = (lib.h2agg_last_error(ctxs[0]) or b"").decode() if ctxs[0] else "RCCL or a device is not available"
            for cx in ctxs:
                if cx:
                    lib.h2agg_destroy(cx)
            raise H2AggError(rc, "h2agg_comm_create: " + msg)
        self._lib, self._ctxs = lib, ctxs
        self.engines = []
        for i, d in enumerate(devices):          # wrap the contexts so the whole H2Agg surface is usable per device
            e = H2Agg.__new__(H2Agg)
            e._lib, e._ctx, e.device = lib, C.c_void_p(ctxs[i]), d
            self.engines.append(e)

    def allgather_add_points(self, partials: Sequence[bytes]) -> bytes:
        """partials[rank] = that rank's npts Jacobian points -> npts affine sums"""
        n = len(self.engines)
        if len(partials) != n:
            raise ValueError("one partial buffer per device")
        npts = len(partials[0]) // 96
        for p in partials:
            _need(p, 96 * npts, "partial")
        out = C.create_string_buffer(max(64 * npts, 1))
        self.engines[0]._check(self._lib.h2agg_allgather_add_points(self._ctxs, n, b"".join(partials), npts, out))
        return out.raw[:64 * npts]

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []


# ---------------------------------------------------------------------------------------------------
# EvaluationQuerySchema mirror (halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs).  The AST lives
# in the C++ host layer of libh2agg.so; these classes only hold node ids, so tests read like the
# reference's: `commit(cq) + evalq(cq)`, `scalar(v) * acc + q`, `proof.w_x.eval(...)`.
class CommitQuery:                       # evaluation.rs:7-12
    def __init__(self, key: str, commitment: Optional[bytes] = None, eval: Optional[bytes] = None):
        self.key, self.commitment, self.eval = key, commitment, eval


class SchemaBuilder:
    """Arena of schema nodes bound to one H2Agg context."""

    def __init__(self, eng: H2Agg):
        self.eng = eng
        self._lib = eng._lib
        self._s = C.c_void_p()
        eng._check(self._lib.h2agg_schema_create(eng._ctx, C.byref(self._s)))
        eng._register_builder(self)

    def close(self):
        if getattr(self, "_s", None) and getattr(self.eng, "_ctx", None):
            self._lib.h2agg_schema_destroy(self._s)
        self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _node(self, fn, *args) -> "EvaluationQuerySchema":
        out = C.c_uint32()
        self.eng._check(fn(self._s, *args, C.byref(out)))
        return EvaluationQuerySchema(self, out.value)

    def commit(self, cq: CommitQuery) -> "EvaluationQuerySchema":        # commit!  evaluation.rs:41-46
        _need(cq.commitment, 64, "commitment")
        return self._node(self._lib.h2agg_schema_node_commitment, cq.key.encode(), cq.commitment)

    def evalq(self, cq: CommitQuery) -> "EvaluationQuerySchema":         # eval!    evaluation.rs:48-53
        _need(cq.eval, 32, "eval")
        return self._node(self._lib.h2agg_schema_node_eval, cq.eval)

    def scalar(self, s: bytes) -> "EvaluationQuerySchema":               # scalar!  evaluation.rs:55-60
        _need(s, 32, "scalar")
        return self._node(self._lib.h2agg_schema_node_scalar, s)

    @staticmethod
    def keys_array(keys: Sequence[str]):
        """the query keys as a C string array, built once and reusable across calls (the reference formats its keys once per
        proof too; re-encoding hundreds of Python strings per call costs more than the device work)"""
        return (C.c_char_p * len(keys))(*[k.encode() for k in keys])

    def evaluation_queries(self, keys, commitments: bytes, evals: bytes, wrap: bool = True):
        """n x EvaluationQuery::new in one call -> list of schema nodes ([C_i] + eval_i).  `keys`: strings, or the array
        from keys_array().  wrap=False returns the raw node-id array (accepted by batch_multi_open): a proof has hundreds
        of queries and one Python object per query costs more than the device work."""
        n = len(keys)
        _need(commitments, 64 * n, "commitments")
        _need(evals, 32 * n, "evals")
        arr = keys if isinstance(keys, C.Array) else self.keys_array(keys)
        out = (C.c_uint32 * n)()
        self.eng._check(self._lib.h2agg_schema_evaluation_queries(self._s, n, arr, commitments, evals, out))
        if not wrap:
            return out
        return [EvaluationQuerySchema(self, out[i]) for i in range(n)]

    def query_set_commitment(self, query_node, point_aff: bytes):
        node = query_node.node if isinstance(query_node, EvaluationQuerySchema) else int(query_node)
        _need(point_aff, 64, "point_aff")
        self.eng._check(self._lib.h2agg_schema_query_set_commitment(self._s, node, point_aff))

    def batch_multi_open(self, key: str, rotations: Sequence[int], points: bytes, query_nodes, w: bytes,
                         v: bytes, u: bytes):
        """multiopen.rs:23-102 in the C++ host layer -> (w_x, w_g) schema nodes."""
        nq = len(rotations)
        _need(points, 32 * nq, "points")
        _need(w, 64 * (len(w) // 64), "w")
        _need(v, 32, "v")
        _need(u, 32, "u")
        if len(query_nodes) != nq:
            raise ValueError("one query node per rotation")
        rot = rotations if isinstance(rotations, C.Array) else (C.c_int32 * nq)(*rotations)
        if isinstance(query_nodes, C.Array):
            qn = query_nodes
        else:
            qn = (C.c_uint32 * nq)(*[q.node if isinstance(q, EvaluationQuerySchema) else int(q) for q in query_nodes])
        wx, wg = C.c_uint32(), C.c_uint32()
        self.eng._check(self._lib.h2agg_schema_batch_multi_open(self._s, key.encode(), nq, rot, points, qn,
                                                                len(w) // 64, w, v, u, C.byref(wx), C.byref(wg)))
        return EvaluationQuerySchema(self, wx.value), EvaluationQuerySchema(self, wg.value)

    def evaluate_multiopen_proof(self, w_x: "EvaluationQuerySchema", w_g: "EvaluationQuerySchema"):
        """verify.rs:705-731 -> (left_aff, right_aff, names)"""
        left, right = C.create_string_buffer(64), C.create_string_buffer(64)
        self.eng._check(self._lib.h2agg_evaluate_multiopen_proof(self._s, w_x.node, w_g.node, left, right))
        return left.raw, right.raw, self.names()

    def evaluate_multiopen_prepare(self, w_x: "EvaluationQuerySchema", w_g: "EvaluationQuerySchema"):
        """the host half of evaluate_multiopen_proof(w_x, w_g) ahead of time (no device work): commitments may still be
        replaced (query_set_commitment) before the evaluation itself"""
        self.eng._check(self._lib.h2agg_evaluate_multiopen_prepare(self._s, w_x.node, w_g.node))

    def names(self):
        need = self._lib.h2agg_schema_names_joined(self._s, None, 0)
        if need == 0:
            return []
        buf = C.create_string_buffer(need)
        self._lib.h2agg_schema_names_joined(self._s, buf, need)
        return buf.raw[:need].decode().split("\n")[:-1]

    def point_list_len(self) -> int:
        return self._lib.h2agg_schema_point_list_len(self._s)


class EvaluationQuerySchema:
    def __init__(self, builder: SchemaBuilder, node: int):
        self.b, self.node = builder, node

    def __add__(self, other):                                            # impl Add  evaluation.rs:62-72
        return self.b._node(self.b._lib.h2agg_schema_node_add, self.node, other.node)

    def __mul__(self, other):                                            # impl Mul  evaluation.rs:74-84
        return self.b._node(self.b._lib.h2agg_schema_node_mul, self.node, other.node)

    def estimate(self) -> int:                                           # evaluation.rs:295-330
        out = C.c_size_t()
        self.b.eng._check(self.b._lib.h2agg_schema_estimate(self.b._s, self.node, C.byref(out)))
        return out.value

    def eval(self):
        """evaluation.rs:172-203 -> (point_jac, scalar or None, names)"""
        jac, sc, has = C.create_string_buffer(96), C.create_string_buffer(32), C.c_int()
        self.b.eng._check(self.b._lib.h2agg_schema_eval(self.b._s, self.node, jac, C.byref(has), sc))
        return jac.raw, (sc.raw if has.value else None), self.b.names()
