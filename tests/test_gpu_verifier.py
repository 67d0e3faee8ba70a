"""GPU: the verifier-params pipeline + aggregation driver of the product (csrc/verifier.inc through h2agg_verify_aggregation)
against the oracle restatement (oracle/verifier.py) on proofs made by the trapdoor prover (tests/toy_prover.py).

The product gets ONLY what the reference's entry point gets — verifying-key descriptions, instance values, transcript
bytes, [s]_2 and [1]_2 — derives every challenge on the device (Poseidon), evaluates gates / permutation / lookup /
vanishing expressions on the device tape, and must (1) reproduce the oracle's final pair and aggregation challenge bit for
bit and (2) ACCEPT under the pairing check, which a verifier that evaluates anything differently from the prover's model
cannot do; tampered inputs must be rejected."""
import importlib

import pytest

import __graft_entry__ as entry
from oracle import bn254 as O
from oracle import pairing as E
from oracle import schema as S
from oracle import verifier as V
from tests.test_pairing_capi import g2b
from tests.test_verifier_pipeline import SHAPES, make_batch

pytestmark = pytest.mark.gpu


def run_product(pkg, eng, setup, circuits, with_pairing=True):
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    table = eng.bases_upload(b"".join(O.aff_to_bytes(p) for p in setup.g_lagrange))
    vks, arg = [], []
    try:
        for c in circuits:
            vk = ver.VerifyingKey(eng, ver.encode_vk(c.cs, O.aff_to_bytes))
            vks.append(vk)
            proofs = []
            for inst, data in c.proofs:
                assert len(inst) == 1                     # one inner proof per transcript
                proofs.append(([b"".join(O.fe_to_bytes(v) for v in col) for col in inst[0]], data))
            arg.append((vk, c.name, table, proofs))
        if with_pairing:
            return ver.verify_aggregation(eng, arg, g2b(setup.s_g2), g2b(setup.g2))
        return ver.verify_aggregation(eng, arg)
    finally:
        for vk in vks:
            vk.close()
        eng.bases_free(table)


@pytest.mark.parametrize("shape_ids,nproofs", [((0,), 1), ((0,), 3), ((1,), 2), ((2,), 2), ((0, 1, 2), 2)])
def test_product_pipeline_matches_oracle_and_is_accepted(eng, pkg, shape_ids, nproofs):
    setup, circuits = make_batch(0x70 + len(shape_ids) * 8 + nproofs, [SHAPES[i] for i in shape_ids], nproofs)
    want_l, want_r, _plain, _commits, want_lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    left, right, lam, ok = run_product(pkg, eng, setup, circuits)
    assert lam == O.fe_to_bytes(want_lam)
    assert left + right == S.final_pair_bytes(want_l, want_r)
    assert ok is True
    assert E.pairing_check([(want_l, setup.s_g2), (want_r, E.g2_neg(setup.g2))])


def test_tampered_inputs_are_rejected(eng, pkg):
    setup, circuits = make_batch(0x7B, [SHAPES[0]], 2)
    inst, data = circuits[0].proofs[1]
    # an evaluation changed
    bad = bytearray(data)
    first_scalar = None
    left0, right0, lam0, ok0 = run_product(pkg, eng, setup, circuits)
    assert ok0 is True
    bad[len(bad) - 32 * 6 + 3] ^= 0x10          # inside the evaluations (4 W points are the last 128 bytes)
    circuits[0].proofs[1] = (inst, bytes(bad))
    l1, r1, lam1, ok1 = run_product(pkg, eng, setup, circuits)
    assert ok1 is False and (l1, r1) != (left0, right0)
    want_l, want_r, *_ = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    assert l1 + r1 == S.final_pair_bytes(want_l, want_r)           # still the SAME (rejected) pair as the reference computes
    # an instance value changed
    inst2 = [[list(col) for col in row] for row in inst]
    inst2[0][0][1] = (inst2[0][0][1] + 5) % O.R
    circuits[0].proofs[1] = (inst2, data)
    _l, _r, _lam, ok2 = run_product(pkg, eng, setup, circuits)
    assert ok2 is False
    # a W point removed: the W count no longer matches the rotation groups (multiopen.rs:48)
    circuits[0].proofs = [(i_, d_[:-32]) for i_, d_ in [(inst, data), (inst, data)]]
    with pytest.raises(pkg.H2AggError) as ei:
        run_product(pkg, eng, setup, circuits)
    assert ei.value.code == pkg.ERR_INVALID
    # a scalar >= r / a point that does not decode
    circuits[0].proofs = [(inst, data)]
    off = len(data) - 32 * 6
    bad = bytearray(data)
    bad[off:off + 32] = O.R.to_bytes(32, "little")
    circuits[0].proofs = [(inst, bytes(bad))]
    with pytest.raises(pkg.H2AggError) as ei:
        run_product(pkg, eng, setup, circuits)
    assert ei.value.code == pkg.ERR_NONCANONICAL
    bad = bytearray(data)
    x = next(v for v in range(2, 60) if pow((v ** 3 + 3) % O.P, (O.P - 1) // 2, O.P) != 1)
    bad[0:32] = x.to_bytes(32, "little")
    circuits[0].proofs = [(inst, bytes(bad))]
    with pytest.raises(pkg.BadPoint):
        run_product(pkg, eng, setup, circuits)


def test_vk_blob_validation(eng, pkg):
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    rng = O.SplitMix64(3)
    from tests import toy_prover as T
    cs = T.make_constraint_system(rng, **SHAPES[0])
    blob = ver.encode_vk(cs, O.aff_to_bytes)
    ver.VerifyingKey(eng, blob).close()
    for bad in (blob[:-4], blob + b"\0\0\0\0", b"XXXX" + blob[4:], blob[:8] + (99).to_bytes(4, "little") + blob[12:]):
        with pytest.raises(pkg.H2AggError):
            ver.VerifyingKey(eng, bad)
