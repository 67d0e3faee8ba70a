"""GPU parity of the proof wire format row: batch point decompression / compression against the oracle's
G1Affine::{from_bytes,to_bytes} restatement, edge cases as the reference's reader sees them
(systems/halo2/transcript.rs:56-79: None -> "invalid point encoding in proof")."""
import json
import os

import pytest

from oracle import bn254 as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_decompress_compress_roundtrip_vs_oracle(eng):
    rng = O.SplitMix64(0x3137)
    pts = [O.scalar_mul(rng.fr(), O.G1) for _ in range(300)] + [O.INF, O.G1, O.neg(O.G1)]
    pts += [O.neg(p) for p in pts[:50]]
    enc = b"".join(O.compress(p) for p in pts)
    aff = b"".join(O.aff_to_bytes(p) for p in pts)
    assert eng.g1_batch_decompress(enc) == aff
    assert eng.g1_batch_compress(aff) == enc
    assert eng.g1_batch_decompress(b"") == b"" and eng.g1_batch_compress(b"") == b""


def test_decompress_invalid_encodings(eng, pkg):
    good = O.compress(O.scalar_mul(12345, O.G1))
    xs = []
    x = 2
    while len(xs) < 5:                                   # x with x^3 + 3 a non-residue
        try:
            O.decompress(x.to_bytes(32, "little"))
        except ValueError:
            xs.append(x)
        x += 1
    bad = [v.to_bytes(32, "little") for v in xs] + [(O.P + 5).to_bytes(32, "little"),          # x >= p
                                                    bytes(31) + bytes([0x80]),                   # x = 0 with the sign bit
                                                    (O.P - 1).to_bytes(32, "little")[:31] + bytes([0xFF])]  # x >= p after masking
    data = good + b"".join(bad) + good + bytes(32)
    with pytest.raises(pkg.BadPoint):
        eng.g1_batch_decompress(data)
    out, ok = eng.g1_batch_decompress(data, with_ok=True)
    want_ok = []
    for i in range(len(data) // 32):
        try:
            p = O.decompress(data[32 * i:32 * i + 32])
            want_ok.append(1)
            assert out[64 * i:64 * i + 64] == O.aff_to_bytes(p)
        except ValueError:
            want_ok.append(0)
            assert out[64 * i:64 * i + 64] == bytes(64)
    assert list(ok) == want_ok == [1] + [0] * len(bad) + [1, 1]
    with pytest.raises(pkg.H2AggError):                  # compress rejects non-canonical coordinates
        eng.g1_batch_compress((O.P + 1).to_bytes(32, "little") + (2).to_bytes(32, "little"))


def test_wire_fixture(eng):
    with open(os.path.join(GOLD, "wire_kats.json")) as f:
        k = json.load(f)
    enc = bytes.fromhex(k["compressed"])
    assert eng.g1_batch_decompress(enc).hex() == k["affine"]
    assert eng.g1_batch_compress(bytes.fromhex(k["affine"])) == enc


def test_decompress_throughput_shape(eng):
    """2^16 points in one call: every lane runs its own 252-squaring chain; results spot-checked + re-compressed"""
    rng = O.SplitMix64(0x3138)
    base = [O.scalar_mul(rng.fr(), O.G1) for _ in range(64)]
    enc = b"".join(O.compress(p) for p in base) * 1024
    out = eng.g1_batch_decompress(enc)
    assert out[:64 * 64] == b"".join(O.aff_to_bytes(p) for p in base) and out[-64 * 64:] == out[:64 * 64]
    assert eng.g1_batch_compress(out) == enc


def test_read_proofs_layout(eng, pkg):
    import importlib
    import __graft_entry__ as entry
    wire = importlib.import_module(entry.PKG_NAME + ".wire")
    rng = O.SplitMix64(0x3139)
    layout = list("PPSPSSPP")
    proofs, want = [], []
    for _ in range(3):
        pts = [O.scalar_mul(rng.fr(), O.G1) for _ in range(layout.count("P"))]
        scs = [rng.fr() for _ in range(layout.count("S"))]
        ip, isc = iter(pts), iter(scs)
        proofs.append(b"".join(O.compress(next(ip)) if t == "P" else O.fe_to_bytes(next(isc)) for t in layout))
        want.append(([O.aff_to_bytes(p) for p in pts], [O.fe_to_bytes(s) for s in scs]))
    assert wire.read_proofs(eng, layout, proofs) == want
    assert wire.read_proofs(eng, layout, []) == []
    with pytest.raises(wire.ProofFormatError):                       # short read
        wire.read_proofs(eng, layout, [proofs[0][:-1]])
    bad_scalar = bytearray(proofs[0])
    bad_scalar[32 * 2:32 * 3] = (O.R + 1).to_bytes(32, "little")
    with pytest.raises(wire.ProofFormatError):                       # Fr::from_repr rejects >= r
        wire.read_proofs(eng, layout, [bytes(bad_scalar)])
    bad_point = bytearray(proofs[1])
    x = 2
    while True:
        try:
            O.decompress(x.to_bytes(32, "little"))
            x += 1
        except ValueError:
            break
    bad_point[0:32] = x.to_bytes(32, "little")
    with pytest.raises(pkg.BadPoint):                                # "invalid point encoding in proof"
        wire.read_proofs(eng, layout, [proofs[0], bytes(bad_point)])
