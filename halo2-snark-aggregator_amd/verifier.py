"""Host-side mirror of the reference's aggregation entry point for the GPU backend.

    verify_aggregation_proofs_in_chip      halo2-snark-aggregator-api/src/systems/halo2/verify.rs:835-942
    calc_verify_circuit_final_pair         halo2-snark-aggregator-circuit/src/verify_circuit.rs:114-201

`encode_vk` serializes what the path reads of a halo2 VerifyingKey / ConstraintSystem (the format documented in
include/h2agg.h at h2agg_vk_create); `verify_aggregation` hands circuits + proofs to libh2agg.so, which replays the
transcripts, evaluates every expression and both multi_exps on the device and (optionally) runs the pairing check.
Nothing here computes: this is marshalling."""
from __future__ import annotations

import ctypes as C
import struct
from typing import Any, List, Optional, Sequence, Tuple

_KIND = {"advice": 0, "fixed": 1, "instance": 2}
_OPS = {"const": 0, "fixed": 1, "advice": 2, "instance": 3, "challenge": 4, "neg": 5, "sum": 6, "product": 7, "scaled": 8}
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _fe(v: int) -> bytes:
    return (v % R_MOD).to_bytes(32, "little")


def encode_expression(e) -> bytes:
    """halo2_proofs::plonk::Expression (nested tuples: ("const", v) ("fixed" | "advice" | "instance", query index)
    ("challenge", index) ("neg", e) ("sum", a, b) ("product", a, b) ("scaled", e, f)) -> postfix bytecode"""
    t = e[0]
    if t == "const":
        return bytes([_OPS[t]]) + _fe(e[1])
    if t in ("fixed", "advice", "instance", "challenge"):
        return bytes([_OPS[t]]) + struct.pack("<I", e[1])
    if t == "neg":
        return encode_expression(e[1]) + bytes([_OPS[t]])
    if t in ("sum", "product"):
        return encode_expression(e[1]) + encode_expression(e[2]) + bytes([_OPS[t]])
    if t == "scaled":
        return encode_expression(e[1]) + bytes([_OPS[t]]) + _fe(e[2])
    raise ValueError("unknown expression node %r (selectors are removed by keygen: expression.rs:33-35)" % (t,))


def _pad4(b: bytes) -> bytes:
    return b + bytes((4 - len(b) % 4) % 4)


def _exprs(es) -> bytes:
    out = struct.pack("<I", len(es))
    for e in es:
        code = encode_expression(e)
        out += struct.pack("<I", len(code)) + _pad4(code)
    return out


def encode_vk(cs, aff_to_bytes) -> bytes:
    """cs: an object with the fields of oracle/verifier.py::ConstraintSystem (the names follow halo2's accessors);
    aff_to_bytes: point -> 64-byte canonical affine encoding"""
    out = struct.pack("<II", 0x4B563248, 1)
    out += struct.pack("<IIIIII", cs.k, cs.num_advice_columns, cs.num_instance_columns, cs.num_challenges, cs.degree,
                       cs.blinding_factors)
    out += _pad4(bytes(cs.advice_column_phase)) + _pad4(bytes(cs.challenge_phase))
    for qs in (cs.advice_queries, cs.instance_queries, cs.fixed_queries):
        out += struct.pack("<I", len(qs)) + b"".join(struct.pack("<Ii", c, r) for c, r in qs)
    out += struct.pack("<I", len(cs.permutation_columns)) + b"".join(
        struct.pack("<II", _KIND[k], i) for k, i in cs.permutation_columns)
    out += struct.pack("<I", len(cs.fixed_commitments)) + b"".join(aff_to_bytes(p) for p in cs.fixed_commitments)
    out += struct.pack("<I", len(cs.permutation_commitments)) + b"".join(aff_to_bytes(p) for p in cs.permutation_commitments)
    out += _fe(cs.vk_scalar)
    out += struct.pack("<I", len(cs.gates)) + b"".join(_exprs(g) for g in cs.gates)
    out += struct.pack("<I", len(cs.lookups))
    for inputs, tables in cs.lookups:
        out += _exprs(inputs) + _exprs(tables)
    return out


class _CircuitProofs(C.Structure):
    _fields_ = [("vk", C.c_void_p), ("name", C.c_char_p), ("g_lagrange", C.c_uint64), ("nproofs", C.c_size_t),
                ("transcripts", C.POINTER(C.c_char_p)), ("transcript_lens", C.POINTER(C.c_size_t)),
                ("instances", C.POINTER(C.c_char_p)), ("instance_lens", C.POINTER(C.c_uint32))]


class VerifyingKey:
    """h2agg_vk: a parsed verifying-key description bound to one engine"""

    def __init__(self, eng, blob: bytes):
        self.eng = eng
        self._lib = eng._lib
        self._vk = C.c_void_p()
        eng._check(self._lib.h2agg_vk_create(eng._ctx, blob, len(blob), C.byref(self._vk)))
        # header words of the accepted blob (include/h2agg.h): magic, version, k, num_advice, num_instance, ...
        self.k, self.num_advice_columns, self.num_instance_columns = (
            int.from_bytes(blob[8 + 4 * i:12 + 4 * i], "little") for i in range(3))

    def close(self):
        if self._vk:
            self._lib.h2agg_vk_destroy(self._vk)
            self._vk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _marshal_circuits(circuits):
    """-> (array of h2agg_circuit_proofs, objects to keep alive)"""
    arr = (_CircuitProofs * len(circuits))()
    keep = []
    for k, (vk, name, g_lagrange, proofs) in enumerate(circuits):
        n = len(proofs)
        tr = (C.c_char_p * max(n, 1))(*[t for _cols, t in proofs])
        tl = (C.c_size_t * max(n, 1))(*[len(t) for _cols, t in proofs])
        inst = (C.c_char_p * max(n, 1))(*[b"".join(cols) for cols, _t in proofs])
        ncol = vk.num_instance_columns
        for cols, _t in proofs:   # the C side indexes instance_lens[i * num_instance_columns + col]: no ragged input reaches it
            if len(cols) != ncol:
                raise ValueError("circuit %r: a proof carries %d instance columns, the verifying key has %d" % (name, len(cols), ncol))
            if any(len(col) % 32 for col in cols):
                raise ValueError("circuit %r: instance columns are 32 bytes per value" % name)
        lens = (C.c_uint32 * max(n * ncol, 1))(*[len(col) // 32 for cols, _t in proofs for col in cols])
        nm = name.encode()
        keep += [tr, tl, inst, lens, nm]
        arr[k] = _CircuitProofs(vk._vk, nm, g_lagrange, n, tr, tl, inst, lens)
    return arr, keep


def verify_aggregation(eng, circuits: Sequence[Tuple[VerifyingKey, str, int, Sequence[Tuple[Sequence[bytes], bytes]]]],
                       s_g2: Optional[bytes] = None, g2: Optional[bytes] = None, with_commits: bool = False):
    """circuits: [(vk, name, g_lagrange_handle, [(instance columns as bytes (32 B per value), transcript bytes), ...])].
    -> (left_aff, right_aff, lambda, pairing_ok or None); with_commits: a fifth element, the advice commitments per proof
    in aggregation order ([[64-byte affine point per advice column] per proof]: `commits` of verify.rs:852-856)"""
    lib = eng._lib
    arr, keep = _marshal_circuits(circuits)
    left, right, lam = C.create_string_buffer(64), C.create_string_buffer(64), C.create_string_buffer(32)
    ok = C.c_int(-1)
    ncommit = [vk.num_advice_columns for vk, _n, _g, proofs in circuits for _p in proofs]
    cap = 64 * sum(ncommit) if with_commits else 0
    adv = C.create_string_buffer(max(cap, 1)) if with_commits else None
    rc = lib.h2agg_verify_aggregation_ex(eng._ctx, C.cast(arr, C.c_void_p), len(circuits), s_g2, g2 if s_g2 is not None else None,
                                         left, right, lam, C.byref(ok) if s_g2 is not None else None, adv, cap)
    eng._check(rc)
    res = (left.raw, right.raw, lam.raw, (bool(ok.value) if s_g2 is not None else None))
    if not with_commits:
        return res
    commits, off = [], 0
    for n in ncommit:
        commits.append([adv.raw[off + 64 * j: off + 64 * (j + 1)] for j in range(n)])
        off += 64 * n
    return res + (commits,)


_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class _Shard(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("total_proofs", C.c_size_t),
                ("global_index", C.POINTER(C.c_uint32)), ("allgather", _ALLGATHER_FN), ("user", C.c_void_p)]


def dist_allgather(dist, device=None):
    """an `allgather(payload: bytes) -> [bytes per rank]` over an initialised torch.distributed group (gloo on CPU tensors,
    nccl = RCCL on `device`)"""
    import torch

    def allgather(payload: bytes):
        mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        if device is not None:
            mine = mine.to(device)
        out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        return [bytes(t.cpu().numpy().tobytes()) for t in out]
    return allgather


def verify_aggregation_sharded(eng, circuits, global_index: Sequence[int], total_proofs: int, rank: int, world: int,
                               allgather=None, s_g2: Optional[bytes] = None, g2: Optional[bytes] = None):
    """h2agg_verify_aggregation_sharded: `circuits` holds THIS rank's proofs (same layout as verify_aggregation),
    global_index[j] = position of the j-th local proof in the aggregation order, total_proofs = N over all ranks.
    allgather: callable(bytes) -> list of `world` byte strings in rank order (the host's transport; dist_allgather wraps
    torch.distributed), or None to use the engine's RCCL communicator (comm_init_rank).  Every rank gets
    (left_aff, right_aff, lambda, pairing_ok or None), equal to verify_aggregation over all N proofs."""
    lib = eng._lib
    arr, keep = _marshal_circuits(circuits)
    nlocal = sum(len(proofs) for _vk, _n, _g, proofs in circuits)
    if len(global_index) != nlocal:
        raise ValueError("global_index must name every local proof (%d given, %d proofs)" % (len(global_index), nlocal))
    gi = (C.c_uint32 * max(nlocal, 1))(*global_index)
    err: List[BaseException] = []

    def cb(_user, send, nbytes, recv):
        try:
            parts = allgather(C.string_at(send, nbytes))
            if len(parts) != world or any(len(p) != nbytes for p in parts):
                raise ValueError("allgather returned %r parts for world %d" % ([len(p) for p in parts], world))
            C.memmove(recv, b"".join(parts), nbytes * world)
            return 0
        except BaseException as e:   # no exception crosses the C ABI
            err.append(e)
            return 1
    fn = _ALLGATHER_FN(cb) if allgather is not None else _ALLGATHER_FN()
    shard = _Shard(rank, world, total_proofs, gi, fn, None)
    left, right, lam = C.create_string_buffer(64), C.create_string_buffer(64), C.create_string_buffer(32)
    ok = C.c_int(-1)
    rc = lib.h2agg_verify_aggregation_sharded(eng._ctx, C.cast(arr, C.c_void_p), len(circuits), C.byref(shard), s_g2,
                                              g2 if s_g2 is not None else None, left, right, lam,
                                              C.byref(ok) if s_g2 is not None else None, None, 0)
    if err:
        raise err[0]
    eng._check(rc)
    return left.raw, right.raw, lam.raw, (bool(ok.value) if s_g2 is not None else None)
