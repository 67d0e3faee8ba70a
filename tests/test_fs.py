"""CPU: output formats (fs.rs:166-195) and final_pair_to_instances (verify_circuit.rs:768-804)."""
import importlib

import __graft_entry__ as entry
from oracle import bn254 as O
from oracle import schema as S


def test_final_pair_to_instances_and_files(tmp_path):
    entry.load_package()
    fs = importlib.import_module(entry.PKG_NAME + ".fs")
    rng = O.SplitMix64(77)
    for _ in range(20):
        left, right = O.scalar_mul(rng.fr(), O.G1), O.scalar_mul(rng.fr(), O.G1)
        inst = [rng.fr() for _ in range(3)]
        got = fs.final_pair_to_instances(O.aff_to_bytes(left), O.aff_to_bytes(right), [O.fe_to_bytes(s) for s in inst])
        want = S.final_pair_to_instances(left, right, inst)
        assert got == [O.fe_to_bytes(w) for w in want]
        # limbs recombine to the coordinates, parity bit is y mod 2
        a, b = (int.from_bytes(g, "little") for g in got[:2])
        assert (a | ((b & ((1 << 136) - 1)) << 136)) == left[0] and (b >> 136) == (left[1] & 1)
    fs.write_verify_circuit_final_pair(str(tmp_path), O.aff_to_bytes(left), O.aff_to_bytes(right),
                                       [O.fe_to_bytes(s) for s in inst])
    l2, r2, i2 = fs.read_verify_circuit_final_pair(str(tmp_path))
    assert l2 == O.aff_to_bytes(left) and r2 == O.aff_to_bytes(right) and i2 == [O.fe_to_bytes(s) for s in inst]
    assert (tmp_path / "verify_circuit_final_pair.data").read_bytes() == S.final_pair_bytes(left, right, inst)
    fs.write_verify_circuit_instance(str(tmp_path), got)
    assert (tmp_path / "verify_circuit_instance.data").stat().st_size == 32 * len(got)
