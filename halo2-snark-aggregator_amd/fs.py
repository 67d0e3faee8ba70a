"""Data formats on the output side of the path: what the reference writes after
`calc_verify_circuit_final_pair`, so that its own `verify_check` / `verify_solidity` can consume the GPU
backend's result.  Pure byte / bit packing — no field arithmetic.

  write_verify_circuit_final_pair   halo2-snark-aggregator-circuit/src/fs.rs:182-195
      W_x.x || W_x.y || W_g.x || W_g.y || instances...   (each 32-byte LE canonical, `to_repr()`)
  write_verify_circuit_instance     halo2-snark-aggregator-circuit/src/fs.rs:166-180
  final_pair_to_instances           halo2-snark-aggregator-circuit/src/verify_circuit.rs:768-804
      the four public inputs of the outer circuit: the base-field coordinates cut into 4 x 68-bit limbs
      (FiveColumnIntegerChipHelper: LIMBS = 4, LIMB_WIDTH = 68, halo2-ecc-circuit-lib/src/five/integer_chip.rs:16-28),
      two limbs per instance, plus the parity of y in bit 136 of the second instance of each point.
"""
from __future__ import annotations

import os
from typing import List, Sequence

LIMBS = 4
LIMB_WIDTH = 68
_LIMB_MASK = (1 << LIMB_WIDTH) - 1


def _limbs_le(coord: bytes) -> List[int]:
    """IntegerChipHelper::w_to_limb_n_le (halo2-ecc-circuit-lib/src/chips/integer_chip.rs:69-88)"""
    v = int.from_bytes(coord, "little")
    out = [(v >> (LIMB_WIDTH * i)) & _LIMB_MASK for i in range(LIMBS - 1)]
    out.append(v >> (LIMB_WIDTH * (LIMBS - 1)))
    return out


def final_pair_to_instances(left_aff: bytes, right_aff: bytes, instances: Sequence[bytes] = ()) -> List[bytes]:
    """-> list of 32-byte LE Fr values (4 + len(instances))."""
    out = []
    for pt in (left_aff, right_aff):
        if pt == bytes(64):
            raise ValueError("identity point: the reference unwraps coordinates() (verify_circuit.rs:775-778)")
        x, y = _limbs_le(pt[:32]), _limbs_le(pt[32:])
        last_bit = (y[0] & 1) << (2 * LIMB_WIDTH)                  # get_last_bit -> limb_modulus_exps[2]
        out.append((x[0] | (x[1] << LIMB_WIDTH)).to_bytes(32, "little"))
        out.append((x[2] + (x[3] << LIMB_WIDTH) + last_bit).to_bytes(32, "little"))
    out.extend(instances)
    return out


def final_pair_bytes(left_aff: bytes, right_aff: bytes, instances: Sequence[bytes] = ()) -> bytes:
    return left_aff + right_aff + b"".join(instances)


def write_verify_circuit_final_pair(folder: str, left_aff: bytes, right_aff: bytes, instances: Sequence[bytes] = ()):
    with open(os.path.join(folder, "verify_circuit_final_pair.data"), "wb") as f:
        f.write(final_pair_bytes(left_aff, right_aff, instances))


def write_verify_circuit_instance(folder: str, instances: Sequence[bytes]):
    with open(os.path.join(folder, "verify_circuit_instance.data"), "wb") as f:
        for s in instances:
            f.write(s)


def read_verify_circuit_final_pair(folder: str):
    data = open(os.path.join(folder, "verify_circuit_final_pair.data"), "rb").read()
    assert len(data) >= 128 and len(data) % 32 == 0
    return data[:64], data[64:128], [data[i:i + 32] for i in range(128, len(data), 32)]


# ---------------------------------------------------------------------------------------------------------------------
# Input side of the path (halo2-snark-aggregator-circuit/src/fs.rs:40-160): the files `verify_run` reads before it calls
# calc_verify_circuit_final_pair.  Byte layouts only; points are decoded by the device kernels / the library.
#
#   sample_circuit_instance_{NAME}{index}.data   fs.rs:88-97;  written at sample_circuit.rs:97-111: every instance value of
#                                                the proof, column after column, 32-byte LE `to_repr()` each
#   sample_circuit_proof_{NAME}{index}.data      fs.rs:99-108; written at sample_circuit.rs:86-95: the transcript bytes
#   sample_circuit_{PARAMS_NAME}.params          fs.rs:40-55;  ParamsKZG::write of halo2_proofs (scroll-dev-1220, unvendored;
#                                                layout recalled from upstream): k as u32 LE, n = 2^k compressed G1 `g`,
#                                                n compressed G1 `g_lagrange`, compressed G2 `g2`, compressed G2 `s_g2`
#   verify_circuit_instance.data                 fs.rs:137-160 (load_instances: one column, `vec![vec![ret]]`)
#   (the .vkey files are NOT read here: VerifyingKey::read re-synthesises the circuit in Rust, fs.rs:68-86; the backend
#    takes the key as the "H2VK" description of include/h2agg.h, produced on the Rust side from the in-memory key)

def target_circuit_instance_name(name: str, index: int) -> str:
    return "sample_circuit_instance_%s%d.data" % (name, index)


def target_circuit_proof_name(name: str, index: int) -> str:
    return "sample_circuit_proof_%s%d.data" % (name, index)


def target_circuit_params_name(params_name: str) -> str:
    return "sample_circuit_%s.params" % params_name


def load_target_circuit_instance(folder: str, name: str, index: int) -> bytes:
    return open(os.path.join(folder, target_circuit_instance_name(name, index)), "rb").read()


def load_target_circuit_proof(folder: str, name: str, index: int) -> bytes:
    return open(os.path.join(folder, target_circuit_proof_name(name, index)), "rb").read()


_R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def load_instances(buf: bytes, column_lens: Sequence[int] = ()) -> List[List[bytes]]:
    """fs.rs:137-152 `load_instances`: whole 32-byte scalars until the buffer runs out (a trailing partial scalar is
    ignored, as `read_exact` failing ends the loop), `from_repr(..).unwrap()` -> ValueError for a value >= r.  Returns the
    instance columns of ONE inner proof: a single column holding everything (`vec![vec![ret]]`) unless `column_lens` says
    how a TargetCircuit::load_instances override cuts it (sdk examples)."""
    vals = [buf[i:i + 32] for i in range(0, len(buf) - len(buf) % 32, 32)]
    for v in vals:
        if int.from_bytes(v, "little") >= _R_MOD:
            raise ValueError("instance scalar is not canonical (from_repr(..).unwrap() panics, fs.rs:149)")
    if not column_lens:
        return [vals]
    if sum(column_lens) != len(vals):
        raise ValueError("column lengths do not add up to the number of instance values")
    out, k = [], 0
    for n in column_lens:
        out.append(vals[k:k + n])
        k += n
    return out


class KzgParams:
    """ParamsKZG as the path uses it: k, the compressed g / g_lagrange tables, g2 and s_g2"""

    def __init__(self, k: int, g: bytes, g_lagrange: bytes, g2: bytes, s_g2: bytes):
        self.k, self.g, self.g_lagrange, self.g2, self.s_g2 = k, g, g_lagrange, g2, s_g2

    @property
    def n(self) -> int:
        return 1 << self.k


def read_params(data: bytes) -> KzgParams:
    """ParamsKZG::read: k (u32 LE) | g: n x 32 B | g_lagrange: n x 32 B | g2: 64 B | s_g2: 64 B"""
    if len(data) < 4:
        raise ValueError("params file too short")
    k = int.from_bytes(data[:4], "little")
    if k > 28:
        raise ValueError("params file: k = %d" % k)
    n = 1 << k
    need = 4 + 64 * n + 128
    if len(data) != need:
        raise ValueError("params file has %d bytes, k = %d needs %d" % (len(data), k, need))
    return KzgParams(k, data[4:4 + 32 * n], data[4 + 32 * n:4 + 64 * n], data[4 + 64 * n:4 + 64 * n + 64], data[4 + 64 * n + 64:])


def write_params(p: KzgParams) -> bytes:
    return p.k.to_bytes(4, "little") + p.g + p.g_lagrange + p.g2 + p.s_g2


def load_target_circuit_params(folder: str, params_name: str) -> KzgParams:
    return read_params(open(os.path.join(folder, target_circuit_params_name(params_name)), "rb").read())


def upload_g_lagrange(eng, params: KzgParams, precompute: bool = True) -> int:
    """params.g_lagrange decoded on the device (one batch-decompression launch for the 2^k points) and left resident as a
    base table: the bases of assign_instance_commitment (verify.rs:623-635).  Returns the table handle."""
    aff = eng.g1_batch_decompress(params.g_lagrange)
    h = eng.bases_upload(aff)
    if precompute:
        try:
            eng.bases_precompute(h)
        except Exception:
            pass                      # tables above ~2^18 points keep the ordinary MSM path
    return h


def pairing_g2(eng, params: KzgParams):
    """(s_g2, g2) in the 128-byte affine form of h2agg_final_pair_check / h2agg_verify_aggregation"""
    both = eng.g2_batch_decompress(params.s_g2 + params.g2)
    return both[:128], both[128:]
