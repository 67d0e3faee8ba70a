"""CPU: the oracle's G1 law and pairing against PUBLIC known-answer vectors (EIP-196 / EIP-197, go-ethereum precompile
tests) — the only externally published vectors the reference's arithmetic is bound to (its Solidity context calls these
precompiles: templates/verifier.sol:5-37,165-276).  They pin oracle/bn254.py's group law and oracle/pairing.py."""
import json
import os

from oracle import bn254 as O
from oracle import pairing as E

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "eip_kats.json")) as f:
    KATS = json.load(f)


def pt(xy):
    x, y = int(xy[0], 16), int(xy[1], 16)
    return O.INF if x == 0 and y == 0 else (x, y)


def test_eip196_ecadd():
    for k in KATS["ecadd"]:
        a, b = pt(k["a"]), pt(k["b"])
        assert O.is_on_curve(a) and O.is_on_curve(b)
        assert O.add(a, b) == pt(k["out"]), k["name"]


def test_eip196_ecmul():
    for k in KATS["ecmul"]:
        p, s = pt(k["p"]), int(k["s"], 16)
        assert O.is_on_curve(p)
        assert O.scalar_mul(s % O.R, p) == pt(k["out"]), k["name"]      # the group has order r: s and s mod r agree


def parse_pairs(words):
    v = [int(w, 16) for w in words]
    pairs = []
    for i in range(0, len(v), 6):
        p = O.INF if v[i] == 0 and v[i + 1] == 0 else (v[i], v[i + 1])
        q = ((v[i + 3], v[i + 2]), (v[i + 5], v[i + 4]))                  # (re, im) pairs from the im-first encoding
        if q == ((0, 0), (0, 0)):
            q = O.INF
        pairs.append((p, q))
    return pairs


def test_g2_generator_is_eip197_p2():
    g = KATS["g2_generator"]
    assert E.G2 == ((int(g["x_re"], 16), int(g["x_im"], 16)), (int(g["y_re"], 16), int(g["y_im"], 16)))
    assert E.g2_on_curve(E.G2)
    assert E.g2_mul(O.R, E.G2) is O.INF


def test_eip197_pairing_vectors():
    for k in KATS["pairing"]:
        pairs = parse_pairs(k["input"])
        for p, q in pairs:
            assert O.is_on_curve(p) and E.g2_on_curve(q)
        assert E.pairing_check(pairs) is k["expect"], k["name"]


def test_bilinearity_and_order():
    e = E.pairing(E.G2, O.G1)
    assert e != E.F12_ONE and E.f12_pow(e, O.R) == E.F12_ONE
    assert E.pairing(E.g2_mul(5, E.G2), O.scalar_mul(7, O.G1)) == E.f12_pow(e, 35)
    # the KZG shape of the reference's check (verify.rs:733-739): e(W, [s]_2) * e(-s*W, [1]_2) = 1
    s = 0x1234567
    w = O.scalar_mul(99, O.G1)
    assert E.pairing_check([(w, E.g2_mul(s, E.G2)), (O.neg(O.scalar_mul(s, w)), E.G2)])
    assert not E.pairing_check([(w, E.g2_mul(s, E.G2)), (O.neg(O.scalar_mul(s + 1, w)), E.G2)])
