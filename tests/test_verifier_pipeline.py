"""CPU: the oracle restatement of the verifier-params pipeline (oracle/verifier.py: build_params, queries, the lagrange /
permutation / lookup / vanishing expressions, the aggregation driver) checked end to end against the ONE external truth
available without a halo2 prover: proofs made by an independent trapdoor prover (tests/toy_prover.py) must satisfy the
reference's pairing check  e(W_x, [s]_2) * e(W_g, -[1]_2) == 1  (verify.rs:733-739), and stop doing so when a byte of the
transcript, an instance value or the aggregation order changes."""
import pytest

from oracle import bn254 as O
from oracle import pairing as E
from oracle import schema as S
from oracle import verifier as V
from tests import toy_prover as T


def test_fr_constants_match_their_definitions():
    assert V.FR_ROOT_OF_UNITY == pow(7, (O.R - 1) >> V.FR_S, O.R)
    assert pow(V.FR_ROOT_OF_UNITY, 1 << V.FR_S, O.R) == 1 and pow(V.FR_ROOT_OF_UNITY, 1 << (V.FR_S - 1), O.R) == O.R - 1
    assert V.FR_DELTA == pow(7, 1 << V.FR_S, O.R)
    w = V.omega_for_k(5)
    assert pow(w, 32, O.R) == 1 and pow(w, 16, O.R) != 1


def make_batch(seed, shapes, proofs_per_circuit):
    rng = O.SplitMix64(seed)
    dlogs = {}
    circuits = []
    setup = T.Setup(5, rng.fr(), 16)
    for ci, shape in enumerate(shapes):
        cs = T.make_constraint_system(rng, dlogs=dlogs, **shape)
        c = V.CircuitProofs("circuit%d" % ci, cs, setup.g_lagrange)
        for i in range(proofs_per_circuit):
            instances = [[[rng.fr() for _ in range(3 + col)] for col in range(cs.num_instance_columns)]]
            c.proofs.append((instances, T.prove(cs, setup, rng, instances, dlogs, "%s_p%d" % (c.name, i))))
        circuits.append(c)
    return setup, circuits


SHAPES = [
    dict(k=5, n_advice=3, n_fixed=2, n_instance=1, n_gates=2, n_lookups=1, degree=4, n_perm_columns=4),
    dict(k=5, n_advice=4, n_fixed=1, n_instance=2, n_gates=1, n_lookups=0, degree=3, n_perm_columns=5,
         phases=(0, 1), n_challenges=2),
    dict(k=5, n_advice=2, n_fixed=1, n_instance=1, n_gates=3, n_lookups=2, degree=5, n_perm_columns=0),
]


@pytest.mark.parametrize("shape_ids,nproofs", [((0,), 1), ((0,), 3), ((1,), 2), ((0, 1, 2), 2)])
def test_trapdoor_proofs_are_accepted_and_tampering_is_rejected(shape_ids, nproofs):
    setup, circuits = make_batch(0x70 + len(shape_ids) * 8 + nproofs, [SHAPES[i] for i in shape_ids], nproofs)
    left, right, plain, commits, lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    assert E.pairing_check([(left, setup.s_g2), (right, E.g2_neg(setup.g2))])
    assert len(commits) == len(shape_ids) * nproofs and len(plain) == sum(
        len(col) for c in circuits for inst, _ in c.proofs for row in inst for col in row)
    # one flipped bit in an evaluation of the first proof
    inst, data = circuits[0].proofs[0]
    npts_before_evals = None
    bad = bytearray(data)
    bad[len(bad) - 32 * 4 - 1] ^= 1          # inside the scalars/points near the end (a W point or an eval)
    circuits[0].proofs[0] = (inst, bytes(bad))
    try:
        l2, r2, *_ = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
        ok = E.pairing_check([(l2, setup.s_g2), (r2, E.g2_neg(setup.g2))])
    except (AssertionError, ValueError, V.P.TranscriptError):
        ok = False                            # e.g. the tampered bytes no longer decode to as many W points as groups
    assert not ok
    # a changed instance value (the instance commitment enters the transcript and the queries)
    inst2 = [[list(col) for col in row] for row in inst]
    inst2[0][0][0] = (inst2[0][0][0] + 1) % O.R
    circuits[0].proofs[0] = (inst2, data)
    l3, r3, *_ = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    assert not E.pairing_check([(l3, setup.s_g2), (r3, E.g2_neg(setup.g2))])
    circuits[0].proofs[0] = (inst, data)
    l4, r4, *_ = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    assert (l4, r4) == (left, right)


def test_single_proof_is_accepted_on_its_own():
    """verify_single_proof_in_chip (verify.rs:779-833): one proof, no aggregation challenge"""
    setup, circuits = make_batch(0x51, [SHAPES[0]], 1)
    c = circuits[0]
    inst, data = c.proofs[0]
    pchip, ctx = S.OracleEccChip(), S.OracleCtx()
    _plain, commitments = V.assign_instance_commitment(pchip, ctx, inst, c.cs, c.g_lagrange)
    t = V.P.PoseidonTranscriptRead(data)
    proof, _adv, vp = V.verify_single_proof_no_eval(t, pchip, ctx, commitments, c.cs, "circuit0_p0")
    left, right, _names = S.evaluate_multiopen_proof(ctx, S.OracleFieldChip(), pchip, proof)
    assert E.pairing_check([(left, setup.s_g2), (right, E.g2_neg(setup.g2))])
    assert len(vp.w) == 4                      # one W per rotation group: 0, +1, -1, -(blinding_factors + 1)
