"""GPU: BASELINE.json configs[2..4] exercised at their own shapes on one GPU (round-1 verdict: configs[3] / [4] were
never run in any form, and bench.py's aggregate leg printed a final pair nothing checked).

  configs[3]  32 proofs sharded 4-per-GPU over 8 ranks     -> 8 shards evaluated one after another on this GPU, folded
                                                              the way the ranks fold after the all-gather
  configs[2]  the composed flow bench.py times              -> instance-column MSMs (batched, fixed-base levels) ->
                                                              to_affine on the device -> query_set_commitment ->
                                                              evaluate_multiopen_proof, P = 347 queries per proof
  configs[4]  16 proofs per GPU, 2^22-point instance MSM    -> the batched MSM at full size through size-independent
                                                              properties (structured scalars, linearity)

Expected values come from the oracle restatement (oracle/schema.py; its multi_exp swapped for the C restatement of the
reference loop so that 16 x 350 commitments take a second, not a minute)."""
import importlib

import numpy as np
import pytest
import torch

import __graft_entry__ as entry
from oracle import bn254 as O, cref
from oracle import schema as S
from tests.test_dist_gloo import make_proofs, reference_final_pair

pytestmark = pytest.mark.gpu


class CEccChip(S.OracleEccChip):
    """MockEccChip with multi_exp = oracle_multi_exp_naive (the reference's loop, mock/arith/ecc.rs:106-129, in C)"""

    def multi_exp(self, ctx, points, scalars):
        ctx.point_list = [O.debug_fmt(p) for p in points]
        n = len(points)
        out = cref.multi_exp_naive(b"".join(O.aff_to_bytes(p) for p in points),
                                   b"".join(O.fe_to_bytes(s) for s in scalars), n)
        return O.aff_from_bytes(out)


def oracle_pair(specs, lam, first_commitments=None):
    """verify.rs:926-938 fold + evaluate_multiopen_proof over the oracle chips for synthetic.ProofSpec data"""
    proofs = []
    for j, sp in enumerate(specs):
        qs = []
        for k in range(sp.nq):
            c = sp.commitments[64 * k:64 * k + 64]
            if k == 0 and first_commitments is not None:
                c = first_commitments[j]
            qs.append(S.evaluation_query(sp.rotations[k], sp.keys[k], O.fe_from_bytes(sp.points[32 * k:32 * k + 32]),
                                         O.aff_from_bytes(c), O.fe_from_bytes(sp.evals[32 * k:32 * k + 32])))
        w = [O.aff_from_bytes(sp.w[64 * i:64 * i + 64]) for i in range(len(sp.w) // 64)]
        proofs.append(S.batch_multi_open_proofs(sp.key, qs, w, O.fe_from_bytes(sp.v), O.fe_from_bytes(sp.u)))
    agg = S.aggregate_fold(proofs, O.fe_from_bytes(lam))
    l, r, names = S.evaluate_multiopen_proof(S.OracleCtx(), S.OracleFieldChip(), CEccChip(), agg)
    return S.final_pair_bytes(l, r), names


def sharded_pair(pkg, eng, specs, lam, world, first_commitments=None):
    """every shard's partial pair on this one GPU, then the fold of aggregate_sharded after its all-gather"""
    agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
    mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
    syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
    backend = agg.GpuBackend(pkg, eng)
    n_total = len(specs)
    lefts, rights = [], []
    for rank in range(world):
        idx = agg.shard_indices(n_total, world, rank)
        b = backend.new_builder()
        proofs = []
        for i in idx:
            p, q0 = syn.build_proof(b, mo.MultiOpenProof, specs[i])
            if first_commitments is not None:
                b.query_set_commitment(q0, first_commitments[i])
            proofs.append(p)
        local = agg.local_weighted_proof(b, proofs, idx, n_total, lam)
        l, r = (agg.IDENTITY_AFF, agg.IDENTITY_AFF) if local is None else backend.evaluate(b, local)
        lefts.append(l)
        rights.append(r)
        b.close()
    return backend.sum_affine(lefts) + backend.sum_affine(rights)


def test_config3_32_proofs_8_shards(eng, pkg):
    """configs[3]: N = 32, world = 8 (4 proofs per shard), exact against the single-process reference semantics"""
    agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
    mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
    backend = agg.GpuBackend(pkg, eng)
    n_total, world = 32, 8
    lam = 0x1234567890ABCDEF1234567890ABCDEF % O.R
    lefts, rights = [], []
    for rank in range(world):
        idx = agg.shard_indices(n_total, world, rank)
        assert len(idx) == 4
        b = backend.new_builder()
        proofs = make_proofs(mo, backend, b, idx, n_total)
        local = agg.local_weighted_proof(b, proofs, idx, n_total, O.fe_to_bytes(lam))
        l, r = backend.evaluate(b, local)
        lefts.append(l)
        rights.append(r)
        b.close()
    got = backend.sum_affine(lefts) + backend.sum_affine(rights)
    assert got == reference_final_pair(n_total, lam)


@pytest.mark.parametrize("n_total,world", [(4, 1), (16, 1), (16, 2), (32, 8)])
def test_evm_shaped_proofs_vs_oracle(eng, pkg, n_total, world):
    """P = 347 queries per proof (300 advice columns: SURVEY.md 8(d) config 3's EVM-like shape), built with the C++ batch
    builders exactly as bench.py does, sharded `world` ways, against the oracle's fold + eval of the same proofs"""
    syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
    pool = syn.point_pool(eng, 0xA66)
    specs, lam = syn.make_proofs(pool, n_total, 300)
    assert specs[0].nq == 347
    want, _names = oracle_pair(specs, lam)
    assert sharded_pair(pkg, eng, specs, lam, world) == want


def test_config4_128_proofs_8_shards_literal_fold(eng, pkg):
    """configs[4]'s literal fold: N = 128 proofs, world = 8, 16 proofs per shard (reduced query count so that the oracle's
    naive multi_exp stays in seconds): the lambda powers lambda^(N-1-i) reach 127, every shard folds its sixteen, the eight
    partial pairs are added — exact against the single-process reference semantics (verify.rs:926-938)"""
    syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
    agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
    pool = syn.point_pool(eng, 0xA66)
    specs, lam = syn.make_proofs(pool, 128, 4)
    assert all(len(agg.shard_indices(128, 8, r)) == 16 for r in range(8))
    want, _names = oracle_pair(specs, lam)
    assert sharded_pair(pkg, eng, specs, lam, 8) == want
    assert sharded_pair(pkg, eng, specs, lam, 1) == want


@pytest.mark.parametrize("fixed_base", [False, True])
def test_composed_flow_instance_commitment_to_final_pair(eng, pkg, fixed_base):
    """bench.py's aggregate leg end to end (verify.rs:835-942 in the reference's order): assign_instance_commitment for
    every proof as ONE batched MSM against a resident g_lagrange table (with and without fixed-base levels), results made
    affine on the device, patched into the schemas with h2agg_schema_query_set_commitment, then the fold and
    evaluate_multiopen_proof — against the oracle, whose instance commitments are the reference's naive loop
    (verify.rs:623-635: scalar_mul_constant + add)."""
    syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
    n_total, n_inst, n_adv = 4, 250, 40
    rng = O.SplitMix64(0xC0F1)
    dev = torch.device("cuda", 0)
    gk = [rng.fr() for _ in range(256)]                                  # "g_lagrange" = gk_i * G, n = 256, l = 6
    gk_np = np.frombuffer(b"".join(O.fe_to_bytes(k) for k in gk), dtype=np.uint8).reshape(256, 32)
    table = eng.bases_generate(torch.from_numpy(gk_np.copy()).to(dev).data_ptr(), 256)
    try:
        if fixed_base:
            eng.bases_precompute(table)
        g_lagrange = eng.bases_download(table, 0, 256)
        inst = [[rng.fr() for _ in range(n_inst)] for _ in range(n_total)]
        inst[1][7] = 0
        inst[2][0] = O.R - 1
        inst_np = np.frombuffer(b"".join(O.fe_to_bytes(s) for row in inst for s in row), dtype=np.uint8)
        d_inst = torch.from_numpy(inst_np.copy()).to(dev)
        d_out = torch.zeros((n_total, 96), dtype=torch.uint8, device=dev)
        eng.g1_msm_device_batch_async(table, d_inst.data_ptr(), n_inst, n_total, d_out.data_ptr())
        aff = eng.g1_batch_to_affine_device(d_out.data_ptr(), n_total)
        commits = [aff[64 * j:64 * j + 64] for j in range(n_total)]
        # the reference's loop, through the C restatement (scalar_mul_constant + add per instance value)
        for j in range(n_total):
            want_c = cref.multi_exp_naive(g_lagrange[:64 * n_inst], b"".join(O.fe_to_bytes(s) for s in inst[j]), n_inst)
            assert commits[j] == want_c, j
            assert commits[j] == eng.g1_batch_to_affine(
                eng.instance_commitment(table, b"".join(O.fe_to_bytes(s) for s in inst[j]), 256 - 6))
        pool = syn.point_pool(eng, 0xA67, 64)
        specs, lam = syn.make_proofs(pool, n_total, n_adv, seed=0xA67)
        want, want_names = oracle_pair(specs, lam, first_commitments=commits)
        assert sharded_pair(pkg, eng, specs, lam, 1, first_commitments=commits) == want
        assert sharded_pair(pkg, eng, specs, lam, 2, first_commitments=commits) == want
        # without the patch the pair differs: the patch is what carries the instance commitment into the result
        assert sharded_pair(pkg, eng, specs, lam, 1) != want
    finally:
        eng.bases_free(table)


def test_config5_sixteen_instance_msms_of_2pow22(eng):
    """configs[4]'s per-GPU share: 16 proofs, each with a 2^22-point instance-column MSM over one resident table, as ONE
    batched call.  Expected values through size-independent structure: proof j's scalars are c_j * s (computed by the
    product's own Fr kernel), so commitment_j = (c_j * sum k_i s_i) * G; plus linearity across the batch."""
    n, batch = 1 << 22, 16
    rng = np.random.Generator(np.random.PCG64(522))
    raw = rng.bytes(64 * n)
    ks = [int.from_bytes(raw[64 * i:64 * i + 64], "little") % O.R for i in range(n)]
    raw = rng.bytes(64 * n)
    ss = [int.from_bytes(raw[64 * i:64 * i + 64], "little") % O.R for i in range(n)]
    total = 0
    for k, s in zip(ks, ss):
        total += k * s
    total %= O.R
    k_bytes = b"".join(k.to_bytes(32, "little") for k in ks)
    s_bytes = b"".join(s.to_bytes(32, "little") for s in ss)
    del ks, ss
    dev = torch.device("cuda", 0)
    d_k = torch.frombuffer(bytearray(k_bytes), dtype=torch.uint8).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    del d_k
    try:
        cs = [(0x9E3779B97F4A7C15 * (j + 1)) % O.R for j in range(batch)]
        cs[3] = 1
        cs[5] = O.R - 1
        d_inst = torch.empty((batch, n, 32), dtype=torch.uint8, device=dev)
        for j, cj in enumerate(cs):
            prod = eng.fr_batch_op(2, s_bytes, O.fe_to_bytes(cj) * n)                  # c_j * s_i, all i
            d_inst[j] = torch.frombuffer(bytearray(prod), dtype=torch.uint8).to(dev).view(n, 32)
        d_out = torch.zeros((batch, 96), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()   # torch filled d_inst on ITS stream; the library reads it on the context's own
        eng.g1_msm_device_batch_async(table, d_inst.data_ptr(), n, batch, d_out.data_ptr())
        aff = eng.g1_batch_to_affine_device(d_out.data_ptr(), batch)
        for j, cj in enumerate(cs):
            assert aff[64 * j:64 * j + 64] == O.aff_to_bytes(O.scalar_mul(cj * total % O.R, O.G1)), j
        # the same MSMs one at a time (unbatched path, different plan) agree with the batch
        one = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_inst[7].data_ptr(), n))
        assert one == aff[64 * 7:64 * 8]
    finally:
        eng.bases_free(table)
