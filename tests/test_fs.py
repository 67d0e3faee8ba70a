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


def _g2_compress(q):
    """halo2curves 0.2.1 G2Affine::to_bytes (recalled): x.c0 || x.c1, parity of y.c0 in bit 7 of the last byte"""
    if q is O.INF:
        return bytes(64)
    b = bytearray(O.fe_to_bytes(q[0][0]) + O.fe_to_bytes(q[0][1]))
    b[63] |= (q[1][0] & 1) << 7
    return bytes(b)


def test_input_side_files_and_params_layout(tmp_path, pkg):
    """fs.rs:40-160: instance / proof / params files; the G2 halves decode through the library (host arithmetic)"""
    import ctypes as C
    from oracle import pairing as E
    entry.load_package()
    fs = importlib.import_module(entry.PKG_NAME + ".fs")
    rng = O.SplitMix64(0xF5)
    vals = [rng.fr() for _ in range(7)]
    (tmp_path / fs.target_circuit_instance_name("simple", 1)).write_bytes(b"".join(O.fe_to_bytes(v) for v in vals) + b"\x01\x02")
    (tmp_path / fs.target_circuit_proof_name("simple", 1)).write_bytes(b"proofbytes")
    buf = fs.load_target_circuit_instance(str(tmp_path), "simple", 1)
    assert fs.load_instances(buf) == [[O.fe_to_bytes(v) for v in vals]]                 # vec![vec![ret]], partial tail ignored
    assert [len(c) for c in fs.load_instances(buf, (3, 4))] == [3, 4]
    assert fs.load_target_circuit_proof(str(tmp_path), "simple", 1) == b"proofbytes"
    import pytest
    with pytest.raises(ValueError):
        fs.load_instances(O.R.to_bytes(32, "little"))
    # params: k | g | g_lagrange | g2 | s_g2
    k, tau = 3, rng.fr()
    g = [O.scalar_mul(pow(tau, i, O.R), O.G1) for i in range(8)]
    gl = [O.scalar_mul(rng.fr(), O.G1) for _ in range(8)]
    s_g2 = E.g2_mul(tau, E.G2)
    p = fs.KzgParams(k, b"".join(O.compress(x) for x in g), b"".join(O.compress(x) for x in gl), _g2_compress(E.G2), _g2_compress(s_g2))
    (tmp_path / fs.target_circuit_params_name("simple")).write_bytes(fs.write_params(p))
    q = fs.load_target_circuit_params(str(tmp_path), "simple")
    assert (q.k, q.n, q.g, q.g_lagrange, q.g2, q.s_g2) == (3, 8, p.g, p.g_lagrange, p.g2, p.s_g2)
    with pytest.raises(ValueError):
        fs.read_params(fs.write_params(p)[:-1])
    lib = pkg.load_library()
    out = C.create_string_buffer(256)
    assert lib.h2agg_g2_batch_decompress(None, q.s_g2 + q.g2, 2, out) == 0
    def g2b(z):
        return b"".join(O.fe_to_bytes(v) for v in (z[0][0], z[0][1], z[1][0], z[1][1]))
    assert out.raw == g2b(s_g2) + g2b(E.G2)
    for s in (5, 1234567, O.R - 2):                                                       # both parities of y.c0 occur
        z = E.g2_mul(s, E.G2)
        assert lib.h2agg_g2_batch_decompress(None, _g2_compress(z), 1, out) == 0 and out.raw[:128] == g2b(z)
    assert lib.h2agg_g2_batch_decompress(None, bytes(64), 1, out) == 0 and out.raw[:128] == bytes(128)
    bad = bytearray(_g2_compress(E.G2))
    bad[0] ^= 1
    assert lib.h2agg_g2_batch_decompress(None, bytes(bad), 1, out) in (pkg.ERR_BAD_POINT, 0)   # x+1 may land on the twist
