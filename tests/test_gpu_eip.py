"""GPU: the PUBLIC known-answer vectors of the EVM precompiles (EIP-196 ecAdd / ecMul, EIP-197 pairing; the contract the
reference's Solidity context is defined by: solidity/templates/verifier.sol:5-37,165-276) through the HIP kernels and the
C ABI — the same vectors that pin the oracle in tests/test_oracle_pairing.py, so HIP and oracle are pinned by external
values, not only by each other.  Also the toolchain probe the round-1 verdict asked for."""
import shutil

import pytest

from oracle import bn254 as O
from tests.test_oracle_pairing import KATS, parse_pairs, pt
from tests.test_pairing_capi import g2b

pytestmark = pytest.mark.gpu
ONE = (1).to_bytes(32, "little")


def jac(p, z=1):
    return O.jac_to_bytes(p, z)


def test_eip196_ecadd_through_hip(eng):
    a = b"".join(jac(pt(k["a"]), 1 + i) for i, k in enumerate(KATS["ecadd"]))
    b = b"".join(jac(pt(k["b"]), 7 + i) for i, k in enumerate(KATS["ecadd"]))
    got = eng.g1_batch_to_affine(eng.g1_batch_add(a, b))
    assert got == b"".join(O.aff_to_bytes(pt(k["out"])) for k in KATS["ecadd"])
    # the same sums through k_g1_sum (the multi-GPU fold)
    for i, k in enumerate(KATS["ecadd"]):
        s = eng.g1_sum(a[96 * i:96 * i + 96] + b[96 * i:96 * i + 96])
        assert eng.g1_batch_to_affine(s) == O.aff_to_bytes(pt(k["out"])), k["name"]


def test_eip196_ecmul_through_hip(eng):
    ks = KATS["ecmul"]
    bases = b"".join(O.aff_to_bytes(pt(k["p"])) for k in ks)
    scalars = b"".join(O.fe_to_bytes(int(k["s"], 16) % O.R) for k in ks)     # order-r group: s and s mod r agree
    want = b"".join(O.aff_to_bytes(pt(k["out"])) for k in ks)
    assert eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(bases, scalars)) == want          # windowed GLV ladder
    for i, k in enumerate(ks):                                                               # Pippenger, n = 1
        if int(k["s"], 16) % O.R == 0:
            continue
        one = eng.g1_msm(bases[64 * i:64 * i + 64], scalars[32 * i:32 * i + 32])
        assert eng.g1_batch_to_affine(one) == want[64 * i:64 * i + 64], k["name"]
    # all of them in one MSM: sum of the outputs
    acc = O.INF
    for k in ks:
        acc = O.add(acc, pt(k["out"]))
    assert eng.g1_batch_to_affine(eng.g1_msm(bases, scalars)) == O.aff_to_bytes(acc)


def test_eip197_pairing_through_the_library(eng):
    for k in KATS["pairing"]:
        pairs = parse_pairs(k["input"])
        g1 = b"".join(O.aff_to_bytes(p) for p, _q in pairs)
        g2 = b"".join(g2b(q) for _p, q in pairs)
        assert eng.pairing_check(g1, g2) is k["expect"], k["name"]


def test_reference_toolchain_probe(record_property):
    """probe for the reference's toolchain on the GPU box instead of asserting its absence.  If a
    Rust toolchain ever appears here, oracle/_ref (the real reference) becomes buildable and "parity unpinned" must be
    revisited — this test then fails on purpose."""
    found = {exe: shutil.which(exe) for exe in ("cargo", "rustc")}
    record_property("reference_toolchain", str(found))
    print("reference toolchain probe:", found)
    assert found == {"cargo": None, "rustc": None}, "a Rust toolchain is present: build oracle/_ref and pin the oracle with it"
