"""GPU vs the committed golden fixtures (tests/golden/*.json, big-integer oracle outputs)."""
import json
import os

import pytest

from oracle import cref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_fr_fixtures(eng):
    kats = load("field_kats.json")["fr"]
    a = b"".join(bytes.fromhex(k["a"]) for k in kats)
    b = b"".join(bytes.fromhex(k["b"]) for k in kats)
    for op, key in ((0, "add"), (1, "sub"), (2, "mul"), (3, "sqr")):
        assert eng.fr_batch_op(op, a, b if op <= 2 else None) == b"".join(bytes.fromhex(k[key]) for k in kats)
    nz = [k for k in kats if k["inv"] is not None]
    assert eng.fr_batch_op(4, b"".join(bytes.fromhex(k["a"]) for k in nz)) == b"".join(bytes.fromhex(k["inv"]) for k in nz)


def test_point_fixtures(eng):
    kats = load("point_kats.json")
    a = b"".join(bytes.fromhex(k["a_jac"]) for k in kats["add"])
    b = b"".join(bytes.fromhex(k["b_jac"]) for k in kats["add"])
    assert eng.g1_batch_to_affine(eng.g1_batch_add(a, b)) == b"".join(bytes.fromhex(k["sum_aff"]) for k in kats["add"])
    assert eng.g1_batch_to_affine(eng.g1_batch_add(a, b, True)) == b"".join(
        bytes.fromhex(k["diff_aff"]) for k in kats["add"])
    sm = kats["scalar_mul"]
    got = eng.g1_batch_scalar_mul(b"".join(bytes.fromhex(k["base_aff"]) for k in sm),
                                  b"".join(bytes.fromhex(k["scalar"]) for k in sm))
    assert eng.g1_batch_to_affine(got) == b"".join(bytes.fromhex(k["out_aff"]) for k in sm)


def test_msm_fixtures(eng):
    for k in load("msm_kats.json"):
        bases, scalars, want = bytes.fromhex(k["bases_aff"]), bytes.fromhex(k["scalars"]), bytes.fromhex(k["out_aff"])
        assert eng.g1_batch_to_affine(eng.g1_msm(bases, scalars)) == want
        assert eng.g1_batch_to_affine(eng.eval_flat(bases, scalars, bytes([1]) * k["n"])) == want
