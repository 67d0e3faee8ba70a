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
