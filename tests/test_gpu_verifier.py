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


@pytest.fixture(params=["device", "host"])
def backend(request, eng):
    """both sponge backends (h2agg_transcript_configure): the device kernel and the host worker threads"""
    eng.transcript_configure(request.param)
    yield request.param
    eng.transcript_configure("auto")


def run_product(pkg, eng, setup, circuits, with_pairing=True, with_commits=False):
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
            return ver.verify_aggregation(eng, arg, g2b(setup.s_g2), g2b(setup.g2), with_commits=with_commits)
        return ver.verify_aggregation(eng, arg, with_commits=with_commits)
    finally:
        for vk in vks:
            vk.close()
        eng.bases_free(table)


@pytest.mark.parametrize("shape_ids,nproofs", [((0,), 1), ((0,), 3), ((1,), 2), ((2,), 2), ((0, 1, 2), 2)])
def test_product_pipeline_matches_oracle_and_is_accepted(eng, pkg, backend, shape_ids, nproofs):
    setup, circuits = make_batch(0x70 + len(shape_ids) * 8 + nproofs, [SHAPES[i] for i in shape_ids], nproofs)
    want_l, want_r, _plain, want_commits, want_lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    left, right, lam, ok, commits = run_product(pkg, eng, setup, circuits, with_commits=True)
    assert lam == O.fe_to_bytes(want_lam)
    assert left + right == S.final_pair_bytes(want_l, want_r)
    assert ok is True
    assert E.pairing_check([(want_l, setup.s_g2), (want_r, E.g2_neg(setup.g2))])
    # the fourth return value of verify_aggregation_proofs_in_chip: every proof's advice commitments (verify.rs:852-856)
    flat_want = [[O.aff_to_bytes(p) for p in per_proof] for per_proof in want_commits]
    assert commits == flat_want


def test_commit_buffer_too_small_is_refused(eng, pkg):
    import ctypes as C
    setup, circuits = make_batch(0x7A, [SHAPES[0]], 1)
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    table = eng.bases_upload(b"".join(O.aff_to_bytes(p) for p in setup.g_lagrange))
    vk = ver.VerifyingKey(eng, ver.encode_vk(circuits[0].cs, O.aff_to_bytes))
    try:
        inst, data = circuits[0].proofs[0]
        cols = [b"".join(O.fe_to_bytes(v) for v in col) for col in inst[0]]
        arr = (ver._CircuitProofs * 1)()
        tr, tl = (C.c_char_p * 1)(data), (C.c_size_t * 1)(len(data))
        ins = (C.c_char_p * 1)(b"".join(cols))
        lens = (C.c_uint32 * len(cols))(*[len(c) // 32 for c in cols])
        nm = circuits[0].name.encode()
        arr[0] = ver._CircuitProofs(vk._vk, nm, table, 1, tr, tl, ins, lens)
        left, right = C.create_string_buffer(64), C.create_string_buffer(64)
        small = C.create_string_buffer(64)
        rc = eng._lib.h2agg_verify_aggregation_ex(eng._ctx, C.cast(arr, C.c_void_p), 1, None, None, left, right, None, None,
                                                  small, 64 * vk.num_advice_columns - 1)
        assert rc == pkg.ERR_INVALID
    finally:
        vk.close()
        eng.bases_free(table)


def test_tampered_inputs_are_rejected(eng, pkg, backend):
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


def test_verify_run_from_the_reference_files(eng, pkg, tmp_path):
    """The SDK's `verify_run` in the pure-calculation context (sdk/src/lib.rs:129-148 -> verify_circuit.rs:898-1009 ->
    calc_verify_circuit_final_pair), from the files the reference's earlier commands leave in --folder-path (fs.rs:40-160)
    to verify_circuit_final_pair.data (fs.rs:182-195): params decoded on the device, proofs replayed, pairing accepted."""
    from tests import toy_prover as T
    from tests.test_fs import _g2_compress
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    fs = importlib.import_module(entry.PKG_NAME + ".fs")
    rng = O.SplitMix64(0xF11E)
    dlogs = {}
    k = 5
    setup = T.Setup(k, rng.fr(), 1 << k)
    cs = T.make_constraint_system(rng, dlogs=dlogs, **SHAPES[0])
    g = [O.scalar_mul(pow(setup.tau, i, O.R), O.G1) for i in range(1 << k)]
    params = fs.KzgParams(k, b"".join(O.compress(p) for p in g), b"".join(O.compress(p) for p in setup.g_lagrange),
                          _g2_compress(setup.g2), _g2_compress(setup.s_g2))
    (tmp_path / fs.target_circuit_params_name("simple")).write_bytes(fs.write_params(params))
    nproofs = 2
    all_inst = []
    for i in range(nproofs):
        inst = [[[rng.fr() for _ in range(4)]]]
        all_inst.append(inst)
        proof = T.prove(cs, setup, rng, inst, dlogs, "simple_p%d" % i)
        (tmp_path / fs.target_circuit_proof_name("simple", i)).write_bytes(proof)
        (tmp_path / fs.target_circuit_instance_name("simple", i)).write_bytes(b"".join(O.fe_to_bytes(v) for v in inst[0][0]))
    # ---- what verify_run does, on the GPU backend
    folder = str(tmp_path)
    p = fs.load_target_circuit_params(folder, "simple")
    table = fs.upload_g_lagrange(eng, p)
    vk = ver.VerifyingKey(eng, ver.encode_vk(cs, O.aff_to_bytes))
    try:
        s_g2, g2 = fs.pairing_g2(eng, p)
        proofs = []
        for i in range(nproofs):
            cols = fs.load_instances(fs.load_target_circuit_instance(folder, "simple", i))      # one column: vec![vec![ret]]
            proofs.append(([b"".join(col) for col in cols], fs.load_target_circuit_proof(folder, "simple", i)))
        left, right, lam, ok = ver.verify_aggregation(eng, [(vk, "simple", table, proofs)], s_g2, g2)
        assert ok is True
        flat_inst = [O.fe_to_bytes(v) for inst in all_inst for v in inst[0][0]]
        fs.write_verify_circuit_final_pair(folder, left, right, flat_inst)
        fs.write_verify_circuit_instance(folder, fs.final_pair_to_instances(left, right, flat_inst))
    finally:
        vk.close()
        eng.bases_free(table)
    # the oracle agrees on every byte of the output file
    circuits = [V.CircuitProofs("simple", cs, setup.g_lagrange, [(all_inst[i], (tmp_path / fs.target_circuit_proof_name("simple", i)).read_bytes())
                                                                  for i in range(nproofs)])]
    wl, wr, plain, _c, _lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    assert (tmp_path / "verify_circuit_final_pair.data").read_bytes() == S.final_pair_bytes(wl, wr, plain)


def test_recorded_aggregation_is_reused_for_new_proofs_of_the_same_shape(eng, pkg, backend):
    """h2agg_verify_aggregation keeps the host-side recording of a call (csrc/verifier.inc `AggPlan`); a later call with the same
    keys, proof counts and proof length only refills the proof scalars, challenges and commitments.  Same keys, different
    proofs every time: every result must still be the oracle's, bit for bit, and accepted by the pairing; a tampered proof on
    a reused recording must give the reference's (rejected) pair; a refused call must not poison the recording."""
    import copy
    import os
    setup, circuits = make_batch(0x91, [SHAPES[0], SHAPES[2]], 4)
    ver = importlib.import_module(entry.PKG_NAME + ".verifier")
    table = eng.bases_upload(b"".join(O.aff_to_bytes(p) for p in setup.g_lagrange))
    vks = [ver.VerifyingKey(eng, ver.encode_vk(c.cs, O.aff_to_bytes)) for c in circuits]

    def subset(sel, patch=None):
        out = []
        for c in circuits:
            c2 = copy.copy(c)
            c2.proofs = [c.proofs[j] for j in sel]
            out.append(c2)
        if patch:
            patch(out)
        return out

    def product(cs):
        arg = []
        for vk, c in zip(vks, cs):
            proofs = [([b"".join(O.fe_to_bytes(v) for v in col) for col in inst[0]], data) for inst, data in c.proofs]
            arg.append((vk, c.name, table, proofs))
        return ver.verify_aggregation(eng, arg, g2b(setup.s_g2), g2b(setup.g2), with_commits=True)

    def check(cs, accept=True):
        want_l, want_r, _plain, want_commits, want_lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), cs)
        left, right, lam, ok, commits = product(cs)
        assert lam == O.fe_to_bytes(want_lam)
        assert left + right == S.final_pair_bytes(want_l, want_r)
        assert ok is accept
        assert commits == [[O.aff_to_bytes(p) for p in per_proof] for per_proof in want_commits]
        return left, right, lam

    try:
        h0, m0, _ = eng.verify_plan_stats()
        check(subset([0, 1]))                        # recorded
        check(subset([2, 3]))                        # reused with two other proofs per circuit
        check(subset([1, 2]))
        h1, m1, _ = eng.verify_plan_stats()
        assert (h1 - h0, m1 - m0) == (2, 1)
        check(subset([0, 1, 2]))                     # another shape: its own recording
        r_a = check(subset([3, 0]))                  # the first shape again
        h2, m2, kept = eng.verify_plan_stats()
        assert (h2 - h0, m2 - m0) == (3, 2) and kept >= 2
        eng.debug_configure("plan_cache", 0)
        try:
            assert check(subset([3, 0])) == r_a      # recorded afresh: the same bytes
            assert eng.verify_plan_stats()[:2] == (h2, m2)
        finally:
            eng.debug_configure("plan_cache", 1)

        def flip_eval(cs):
            inst, data = cs[0].proofs[1]
            bad = bytearray(data)
            bad[len(bad) - 32 * 6 + 3] ^= 0x10
            cs[0].proofs[1] = (inst, bytes(bad))
        check(subset([0, 1], flip_eval), accept=False)   # on the reused recording: the reference's rejected pair

        def big_scalar(cs):
            inst, data = cs[0].proofs[0]
            bad = bytearray(data)
            off = len(bad) - 32 * 6
            bad[off:off + 32] = O.R.to_bytes(32, "little")
            cs[0].proofs[0] = (inst, bytes(bad))
        with pytest.raises(pkg.H2AggError) as ei:
            product(subset([2, 3], big_scalar))
        assert ei.value.code == pkg.ERR_NONCANONICAL
        check(subset([2, 3]))                        # and the recording still serves good proofs
        # a key made again from the same description is a different key: no reuse across keys
        vks[0].close()
        vks[0] = ver.VerifyingKey(eng, ver.encode_vk(circuits[0].cs, O.aff_to_bytes))
        h3, m3, _ = eng.verify_plan_stats()
        check(subset([0, 1]))
        assert eng.verify_plan_stats()[1] == m3 + 1
    finally:
        for vk in vks:
            vk.close()
        eng.bases_free(table)


def test_phase_split_prewake_and_tape_keys_do_not_change_the_result(eng, pkg):
    """round 6's latency items are switches over the SAME computation: h2agg_last_phases (debug key phases) reports the call's
    wall-clock split and per-chain stamps; prewake = 0 keeps the sponge workers asleep until their chains are posted and the
    pairing's two Miller loops on one thread; tape_lds = 0 runs the Fr tape through the L2 register file.  Every combination
    must return the oracle's pair, lambda and an accepted pairing; a tampered proof must still be rejected under each."""
    setup, circuits = make_batch(0xA7, [SHAPES[0], SHAPES[1]], 2)
    want_l, want_r, _plain, _c, want_lam = V.verify_aggregation_proofs_in_chip(S.OracleEccChip(), circuits)
    want = (S.final_pair_bytes(want_l, want_r), O.fe_to_bytes(want_lam))
    eng.transcript_configure("host")
    try:
        assert eng.last_phases() == "" or "=" in eng.last_phases()
        eng.debug_configure("phases", 1)
        for prewake in (1, 0):
            for tape_lds in (1, 0):
                eng.debug_configure("prewake", prewake)
                eng.debug_configure("tape_lds", tape_lds)
                for _ in range(3):                               # recorded, then reused twice
                    left, right, lam, ok = run_product(pkg, eng, setup, circuits)
                    assert (left + right, lam) == want and ok is True, (prewake, tape_lds)
                    line = eng.last_phases()
                    for name in ("inst_upload=", "sponge_wait=", "evaluate=", "pairing=", "sponge chains posted at", "caller cpu"):
                        assert name in line, (name, line)
                    assert ("register file in LDS" in line) == bool(tape_lds) or "tape:" not in line
        # a one-bit change of a proof under the default switches: rejected, and the phase split is that call's
        eng.debug_configure("prewake", 1)
        eng.debug_configure("tape_lds", 1)
        import copy
        c2 = copy.copy(circuits[0])
        inst, data = c2.proofs[0]
        pos = len(data) - 40                                     # inside the last scalars: decodes, verifies to a different pair
        c2.proofs = [(inst, data[:pos] + bytes([data[pos] ^ 1]) + data[pos + 1:])] + list(c2.proofs[1:])
        try:
            _l, _r, _lam, ok = run_product(pkg, eng, setup, [c2, circuits[1]])
            assert ok is False
        except pkg.H2AggError:
            pass                                                 # (a non-canonical scalar is refused outright)
    finally:
        eng.debug_configure("phases", 0)
        eng.debug_configure("prewake", 1)
        eng.debug_configure("tape_lds", 1)
        eng.transcript_configure("auto")
    assert eng.last_phases() != ""
