"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Restatement of the verifier-params pipeline that produces the hot path's inputs and of the
aggregation driver around it:

  VerifierParamsBuilder::build_params       halo2-snark-aggregator-api/src/systems/halo2/verify.rs:342-571
      init_transcript :57-75, squeeze_instance_commitment :77-97, build_permutation_evaluated :206-292,
      build_lookup_evaluated :294-340, rotate_omega :158-173
  VerifierParams::queries                   .../params.rs:74-224
  LagrangeGenerator::get_lagrange_commits   .../lagrange.rs:17-39
  Evaluable::chip_evaluate                  .../expression.rs:19-113
  permutation::Evaluated::{expressions,queries}, CommonEvaluated::queries     .../permutation.rs:34-201
  lookup::Evaluated::{expressions,queries}  .../lookup.rs:34-182
  vanish::Evaluated::{new,queries}          .../vanish.rs:18-92
  assign_instance_commitment                .../verify.rs:574-649
  verify_single_proof_no_eval               .../verify.rs:651-688
  verify_aggregation_proofs_in_chip         .../verify.rs:835-942
  calc_verify_circuit_final_pair            halo2-snark-aggregator-circuit/src/verify_circuit.rs:114-201

The reference drives all of this from a halo2_proofs `VerifyingKey` / `ConstraintSystem` (unvendored).  What the path
reads of it is restated as a plain description, `ConstraintSystem` below, field by field with the accessor it stands for.
Fr arithmetic is plain integers mod r (exact, so the order of chip calls is immaterial); group operations go through a
`pchip` object with the MockEccChip methods so that the same code also runs "in the exponent" (tests/toy_prover.py).

PARITY PINNING STATUS: "parity unpinned" — no real halo2 proof or VerifyingKey exists in this image.  Pinned facts: Fr's
2^28-th root of unity and DELTA are checked against their definitions (7^((r-1)/2^28), 7^(2^28)); the control flow is read
against the cited lines; tests/test_verifier_pipeline.py checks that proofs built by an independent trapdoor prover are
ACCEPTED by the pairing check (and rejected after a one-byte change), which no self-consistent but wrong verifier would do.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple

from . import bn254 as O
from . import schema as S
from . import poseidon as P

R = O.R
# halo2curves bn256::Fr constants (unvendored; checked against their definitions in tests/test_verifier_pipeline.py)
FR_S = 28
FR_ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C      # 7^((r-1)/2^28)
FR_DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2               # 7^(2^28)


def omega_for_k(k: int) -> int:
    """EvaluationDomain::get_omega(): ROOT_OF_UNITY^(2^(S - k))"""
    w = FR_ROOT_OF_UNITY
    for _ in range(FR_S - k):
        w = w * w % R
    return w


# ---------------------------------------------------------------------------------------------- description of a VK
# Expression<F> (halo2_proofs::plonk::Expression) as nested tuples:
#   ("const", v) ("fixed", query_index) ("advice", query_index) ("instance", query_index) ("challenge", index)
#   ("neg", e) ("sum", a, b) ("product", a, b) ("scaled", e, f)          (Selector is gone after keygen: expression.rs:33-35)
@dataclass
class ConstraintSystem:
    """what build_params / queries read of `vk.cs()`, `vk.get_domain()`, `vk.fixed_commitments()`, `vk.permutation()`"""
    k: int                                              # params.k; n = 2^k (verify.rs:470)
    num_advice_columns: int                             # cs.num_advice_columns()
    num_instance_columns: int                           # cs.num_instance_columns
    num_challenges: int                                 # cs.num_challenges()
    advice_column_phase: List[int]                      # cs.advice_column_phase
    challenge_phase: List[int]                          # cs.challenge_phase
    advice_queries: List[Tuple[int, int]]               # cs.advice_queries   (column index, rotation)
    instance_queries: List[Tuple[int, int]]             # cs.instance_queries
    fixed_queries: List[Tuple[int, int]]                # cs.fixed_queries
    gates: List[List[Any]]                              # cs.gates[i].polys
    lookups: List[Tuple[List[Any], List[Any]]]          # cs.lookups[j].(input_expressions, table_expressions)
    permutation_columns: List[Tuple[str, int]]          # cs.permutation.columns: ("advice" | "fixed" | "instance", index)
    degree: int                                         # cs.degree()
    blinding_factors: int                               # cs.blinding_factors()
    fixed_commitments: List[Any]                        # vk.fixed_commitments()
    permutation_commitments: List[Any]                  # vk.permutation().commitments
    vk_scalar: int = 0                                  # from_bytes_wide(blake2b("Halo2-Verify-Key", pinned vk)) (verify.rs:57-70)

    @property
    def n(self) -> int:
        return 1 << self.k

    @property
    def omega(self) -> int:
        return omega_for_k(self.k)

    def phases(self) -> List[int]:                      # cs.phases(): 0 ..= max phase
        top = max(list(self.advice_column_phase) + list(self.challenge_phase) + [0])
        return list(range(top + 1))

    @property
    def chunk_len(self) -> int:
        return self.degree - 2

    @property
    def num_permutation_sets(self) -> int:              # permutation.columns.chunks(degree - 2).len()
        c = self.chunk_len
        return (len(self.permutation_columns) + c - 1) // c

    @property
    def quotient_poly_degree(self) -> int:              # domain.get_quotient_poly_degree() = degree - 1
        return self.degree - 1

    def any_query_index(self, kind: str, column: int) -> int:
        """cs.get_any_query_index(column, Rotation::cur())"""
        qs = {"advice": self.advice_queries, "fixed": self.fixed_queries, "instance": self.instance_queries}[kind]
        return qs.index((column, 0))


# ---------------------------------------------------------------------------------------------- expression.rs
def evaluate_expression(e, fixed, advice, instance, challenges) -> int:
    t = e[0]
    if t == "const":
        return e[1] % R
    if t == "fixed":
        return fixed[e[1]]
    if t == "advice":
        return advice[e[1]]
    if t == "instance":
        return instance[e[1]]
    if t == "challenge":
        return challenges[e[1]]
    if t == "neg":
        return (-evaluate_expression(e[1], fixed, advice, instance, challenges)) % R
    if t == "sum":
        return (evaluate_expression(e[1], fixed, advice, instance, challenges) +
                evaluate_expression(e[2], fixed, advice, instance, challenges)) % R
    if t == "product":
        return evaluate_expression(e[1], fixed, advice, instance, challenges) * \
            evaluate_expression(e[2], fixed, advice, instance, challenges) % R
    if t == "scaled":
        return e[2] % R * evaluate_expression(e[1], fixed, advice, instance, challenges) % R
    raise ValueError("virtual selectors are removed during optimization (expression.rs:33-35)")


def mul_add_accumulate(vals: Sequence[int], b: int) -> int:
    """ArithFieldChip::mul_add_accumulate (arith/field.rs:68-81): Horner, acc = acc*b + v"""
    acc = 0
    for v in vals:
        acc = (acc * b + v) % R
    return acc


# ---------------------------------------------------------------------------------------------- assign_instance_commitment
def assign_instance_commitment(pchip, ctx, instances, cs: ConstraintSystem, g_lagrange):
    """verify.rs:574-649 -> (plain_assigned_instances, commitments[proof][column])"""
    plain = []
    for inst in instances:
        assert len(inst) == cs.num_instance_columns                                    # :591-593
    commitments = []
    for inst in instances:
        row = []
        for column in inst:
            assert len(column) <= cs.n - (cs.blinding_factors + 1)                     # :600-603
            plain.extend(v % R for v in column)
            acc = None
            for i, v in enumerate(column):
                ls = pchip.scalar_mul_constant(ctx, v % R, g_lagrange[i])
                acc = ls if acc is None else pchip.add(ctx, acc, ls)
            row.append(pchip.assign_const(ctx, O.INF) if acc is None else pchip.normalize(ctx, acc))
        commitments.append(row)
    return plain, commitments


# ---------------------------------------------------------------------------------------------- build_params
@dataclass
class VerifierParams:
    key: str
    cs: ConstraintSystem
    instance_commitments: list
    instance_evals: list
    challenges: list
    advice_commitments: list
    advice_evals: list
    fixed_evals: list
    permutation_evals: list
    permutation_sets: list          # [proof][set] = (commitment, eval, next_eval, last_eval or None)
    lookups: list                   # [proof][j] = dict
    vanish_commitments: list
    random_commitment: Any
    random_eval: int
    w: list
    theta: int
    beta: int
    gamma: int
    y: int
    x: int
    v: int
    u: int
    x_next: int = 0
    x_last: int = 0
    x_inv: int = 0
    xn: int = 0


def rotate_omega(x: int, omega: int, at: int) -> int:
    base = O.inv(omega, R) if at < 0 else omega
    return x * pow(base, abs(at), R) % R


def build_params(transcript, pchip, ctx, assigned_instances, cs: ConstraintSystem, key: str) -> VerifierParams:
    """verify.rs:342-571, the transcript replay in the reference's order"""
    t = transcript
    t.common_scalar(cs.vk_scalar % R)                                                  # init_transcript :57-75
    for inst in assigned_instances:                                                    # squeeze_instance_commitment
        for p in inst:
            t.common_point(pchip.to_value(p))
    num_proofs = len(assigned_instances)
    advice = [[None] * cs.num_advice_columns for _ in range(num_proofs)]
    challenges = [0] * cs.num_challenges
    for phase in cs.phases():                                                          # :360-377
        for row in advice:
            for col, ph in enumerate(cs.advice_column_phase):
                if ph == phase:
                    row[col] = t.read_point()
        for idx, ph in enumerate(cs.challenge_phase):
            if ph == phase:
                challenges[idx] = t.squeeze_challenge_scalar()
    theta = t.squeeze_challenge_scalar()
    lookups_permuted = [[(t.read_point(), t.read_point()) for _ in cs.lookups] for _ in range(num_proofs)]
    beta = t.squeeze_challenge_scalar()
    gamma = t.squeeze_challenge_scalar()
    perms_committed = [[t.read_point() for _ in range(cs.num_permutation_sets)] for _ in range(num_proofs)]
    lookups_committed = [[t.read_point() for _ in cs.lookups] for _ in range(num_proofs)]
    random_commitment = t.read_point()
    y = t.squeeze_challenge_scalar()
    h_commitments = [t.read_point() for _ in range(cs.quotient_poly_degree)]
    x = t.squeeze_challenge_scalar()
    instance_evals = [[t.read_scalar() for _ in cs.instance_queries] for _ in range(num_proofs)]
    advice_evals = [[t.read_scalar() for _ in cs.advice_queries] for _ in range(num_proofs)]
    fixed_evals = [t.read_scalar() for _ in cs.fixed_queries]
    random_eval = t.read_scalar()
    permutation_evals = [t.read_scalar() for _ in cs.permutation_commitments]
    permutation_sets = []                                                              # build_permutation_evaluated
    for committed in perms_committed:
        sets = []
        for i, commitment in enumerate(committed):
            ev = t.read_scalar()
            nx = t.read_scalar()
            last = t.read_scalar() if i + 1 < len(committed) else None
            sets.append((commitment, ev, nx, last))
        permutation_sets.append(sets)
    lookups = []                                                                       # build_lookup_evaluated
    for i in range(num_proofs):
        row = []
        for j, (inputs, tables) in enumerate(cs.lookups):
            row.append({
                "product_eval": t.read_scalar(), "product_next_eval": t.read_scalar(),
                "permuted_input_eval": t.read_scalar(), "permuted_input_inv_eval": t.read_scalar(),
                "permuted_table_eval": t.read_scalar(),
                "permuted_input_commitment": lookups_permuted[i][j][0],
                "permuted_table_commitment": lookups_permuted[i][j][1],
                "product_commitment": lookups_committed[i][j],
                "input_expressions": inputs, "table_expressions": tables,
                "key": "%s_%d_%d" % (key, i, j)})
        lookups.append(row)
    v = t.squeeze_challenge_scalar()
    w = []
    while True:                                                                        # `while let Ok(p) = self.load_point()`
        try:
            w.append(t.read_point())
        except (P.TranscriptError, ValueError):
            break
    u = t.squeeze_challenge_scalar()
    omega = cs.omega
    l = cs.blinding_factors + 1
    return VerifierParams(
        key=key, cs=cs, instance_commitments=assigned_instances, instance_evals=instance_evals, challenges=challenges,
        advice_commitments=advice, advice_evals=advice_evals, fixed_evals=fixed_evals, permutation_evals=permutation_evals,
        permutation_sets=permutation_sets, lookups=lookups, vanish_commitments=h_commitments,
        random_commitment=random_commitment, random_eval=random_eval, w=w, theta=theta, beta=beta, gamma=gamma, y=y, x=x,
        v=v, u=u, x_next=rotate_omega(x, omega, 1), x_last=rotate_omega(x, omega, -l), x_inv=rotate_omega(x, omega, -1),
        xn=pow(x, cs.n, R))


# ---------------------------------------------------------------------------------------------- queries
def lagrange_commits(vp: VerifierParams) -> List[int]:
    """lagrange.rs:17-39: l_i(x) = (w_i / n) (x^n - 1) / (x - w_i), w_i = omega^-i, i = 0 ..= l"""
    cs = vp.cs
    l = cs.blinding_factors + 1
    omega_inv = O.inv(cs.omega, R)
    ws = [1]
    for _ in range(l):
        ws.append(ws[-1] * omega_inv % R)
    n = cs.n % R
    return [wi * O.inv(n, R) % R * ((vp.xn - 1) % R) % R * O.inv((vp.x - wi) % R, R) % R for wi in ws]


def permutation_expressions(vp: VerifierParams, k: int, l_0, l_last, l_blind) -> List[int]:
    """permutation.rs:54-136"""
    cs = vp.cs
    sets = vp.permutation_sets[k]
    # permutation_evaluated_evals (verify.rs:240-268): the column evals at Rotation::cur(), chunked
    evals = []
    for kind, col in cs.permutation_columns:
        q = cs.any_query_index(kind, col)
        evals.append({"advice": vp.advice_evals[k], "fixed": vp.fixed_evals, "instance": vp.instance_evals[k]}[kind][q])
    res = []
    if sets:
        res.append(l_0 * ((1 - sets[0][1]) % R) % R)
        z = sets[-1][1]
        res.append(l_last * ((z * z - z) % R) % R)
    for i in range(1, len(sets)):
        res.append((sets[i][1] - sets[i - 1][3]) % R * l_0 % R)
    t0 = vp.beta * vp.x % R
    t1 = (1 - (l_last + l_blind)) % R
    c = cs.chunk_len
    for chunk_index, st in enumerate(sets):
        left, right = st[2], st[1]
        delta_pow = 1 if chunk_index == 0 else pow(FR_DELTA, chunk_index * c, R)
        d = t0 * delta_pow % R
        for ev, pev in zip(evals[chunk_index * c:(chunk_index + 1) * c],
                           vp.permutation_evals[chunk_index * c:(chunk_index + 1) * c]):
            t2 = (ev + vp.gamma) % R
            left = (t2 + vp.beta * pev) % R * left % R
            right = (t2 + d) % R * right % R
            d = FR_DELTA * d % R
        res.append((left - right) % R * t1 % R)
    return res


def lookup_expressions(vp: VerifierParams, k: int, lk: dict, l_0, l_last, l_blind) -> List[int]:
    """lookup.rs:34-119"""
    z_wx, z_x = lk["product_next_eval"], lk["product_eval"]
    a_x, s_x, a_invwx = lk["permuted_input_eval"], lk["permuted_table_eval"], lk["permuted_input_inv_eval"]
    ev = lambda e: evaluate_expression(e, vp.fixed_evals, vp.advice_evals[k], vp.instance_evals[k], vp.challenges)
    left = z_wx * ((a_x + vp.beta) % R) % R * ((s_x + vp.gamma) % R) % R
    input_eval = mul_add_accumulate([ev(e) for e in lk["input_expressions"]], vp.theta)
    table_eval = mul_add_accumulate([ev(e) for e in lk["table_expressions"]], vp.theta)
    t0 = (1 - (l_last + l_blind)) % R
    t1 = (a_x - s_x) % R
    return [
        l_0 * ((1 - z_x) % R) % R,
        l_last * ((z_x * z_x - z_x) % R) % R,
        (left - z_x * ((input_eval + vp.beta) % R) % R * ((table_eval + vp.gamma) % R)) % R * t0 % R,
        l_0 * t1 % R,
        t1 * ((a_x - a_invwx) % R) % R * t0 % R,
    ]


def queries(vp: VerifierParams) -> list:
    """VerifierParams::queries (params.rs:74-224) -> [S.EvaluationQuery-like (rotation, point, schema)]"""
    cs = vp.cs
    ls = lagrange_commits(vp)
    l = cs.blinding_factors + 1
    l_0, l_last = ls[0], ls[l]
    l_blind = sum(ls[1:l]) % R
    expression = []
    for k in range(len(vp.advice_evals)):
        for gate in cs.gates:
            for poly in gate:
                expression.append(evaluate_expression(poly, vp.fixed_evals, vp.advice_evals[k], vp.instance_evals[k],
                                                      vp.challenges))
        expression += permutation_expressions(vp, k, l_0, l_last, l_blind)
        for lk in vp.lookups[k]:
            expression += lookup_expressions(vp, k, lk, l_0, l_last, l_blind)
    omega = cs.omega
    qs = []
    for i in range(len(vp.instance_commitments)):
        for qi, (column, at) in enumerate(cs.instance_queries):
            qs.append(S.evaluation_query(at, "%s_instance_commitments%d" % (vp.key, column), rotate_omega(vp.x, omega, at),
                                         vp.instance_commitments[i][column], vp.instance_evals[i][qi]))
        for qi, (column, at) in enumerate(cs.advice_queries):
            qs.append(S.evaluation_query(at, "%s_advice_commitments%d" % (vp.key, column), rotate_omega(vp.x, omega, at),
                                         vp.advice_commitments[i][column], vp.advice_evals[i][qi]))
        pkey = "%s_%d" % (vp.key, i)                                                   # verify.rs:286
        sets = vp.permutation_sets[i]
        for si, st in enumerate(sets):                                                 # permutation.rs:138-201
            name = "%s_permutation_product_commitment_%d" % (pkey, si)
            qs.append(S.evaluation_query(0, name, vp.x, st[0], st[1]))
            qs.append(S.evaluation_query(1, name, vp.x_next, st[0], st[2]))
        for si in range(len(sets) - 2, -1, -1):                                        # .rev().skip(1)
            st = sets[si]
            qs.append(S.evaluation_query(-(cs.blinding_factors + 1), "%s_permutation_product_commitment_%d" % (pkey, si),
                                         vp.x_last, st[0], st[3]))
        for lk in vp.lookups[i]:                                                       # lookup.rs:121-181
            kk = lk["key"]
            qs.append(S.evaluation_query(0, kk + "_product_commitment", vp.x, lk["product_commitment"], lk["product_eval"]))
            qs.append(S.evaluation_query(0, kk + "_permuted_input_commitment", vp.x, lk["permuted_input_commitment"],
                                         lk["permuted_input_eval"]))
            qs.append(S.evaluation_query(0, kk + "_permuted_table_commitment", vp.x, lk["permuted_table_commitment"],
                                         lk["permuted_table_eval"]))
            qs.append(S.evaluation_query(-1, kk + "_permuted_input_commitment", vp.x_inv, lk["permuted_input_commitment"],
                                         lk["permuted_input_inv_eval"]))
            qs.append(S.evaluation_query(1, kk + "_product_commitment", vp.x_next, lk["product_commitment"],
                                         lk["product_next_eval"]))
    for qi, (column, at) in enumerate(cs.fixed_queries):
        qs.append(S.evaluation_query(at, "%s_fixed_commitments%d" % (vp.key, column), rotate_omega(vp.x, omega, at),
                                     cs.fixed_commitments[column], vp.fixed_evals[qi]))
    for i, (commitment, ev) in enumerate(zip(cs.permutation_commitments, vp.permutation_evals)):   # permutation.rs:34-52
        qs.append(S.evaluation_query(0, "%s_permutation_commitments%d" % (vp.key, i), vp.x, commitment, ev))
    # vanish.rs:18-72
    expected_h_eval = mul_add_accumulate(expression, vp.y) * O.inv((vp.xn - 1) % R, R) % R
    h = None
    for i, c in enumerate(reversed(vp.vanish_commitments)):
        cq = S.commit(S.CommitQuery("%s_h_commitment%d" % (vp.key, i), c, None))
        h = cq if h is None else S.scalar(vp.xn) * h + cq
    qs.append((0, vp.x, h + S.scalar(expected_h_eval)))
    qs.append(S.evaluation_query(0, "%s_random_commitment" % vp.key, vp.x, vp.random_commitment, vp.random_eval))
    return qs


# ---------------------------------------------------------------------------------------------- drivers
def verify_single_proof_no_eval(transcript, pchip, ctx, assigned_instances, cs, key):
    """verify.rs:651-688 -> (MultiOpenProof, advice_commitments[0])"""
    vp = build_params(transcript, pchip, ctx, assigned_instances, cs, key)
    return S.batch_multi_open_proofs(vp.key, queries(vp), vp.w, vp.v, vp.u), vp.advice_commitments[0], vp


@dataclass
class CircuitProofs:
    """CircuitProof (verify.rs:769-783): one circuit (vk + params) with its proofs"""
    name: str
    cs: ConstraintSystem
    g_lagrange: list
    proofs: List[Tuple[list, bytes]] = field(default_factory=list)      # (instances [inner proof][column][values], transcript)


def verify_aggregation_proofs_in_chip(pchip, circuits: List[CircuitProofs], ctx=None, make_transcript=None):
    """verify.rs:835-942 with the transcripts of calc_verify_circuit_final_pair (verify_circuit.rs:121-163): one Poseidon
    reader per proof over its bytes, an empty one for the aggregation challenge.  Returns (left, right, instances,
    advice commitments, lambda)."""
    ctx = ctx if ctx is not None else S.OracleCtx()
    mk = make_transcript or (lambda data: P.PoseidonTranscriptRead(data))
    main = mk(b"")
    plain = []
    proofs = []
    for ci, circuit in enumerate(circuits):
        transcripts = []
        for i, (instances, data) in enumerate(circuit.proofs):
            t = mk(data)
            transcripts.append(t)
            assigned, commitments = assign_instance_commitment(pchip, ctx, instances, circuit.cs, circuit.g_lagrange)
            plain += assigned
            key = "%s_p%d" % (circuit.name, i)                                         # verify_circuit.rs:140
            p, c, _vp = verify_single_proof_no_eval(t, pchip, ctx, commitments, circuit.cs, key)
            proofs.append((p, c))
        for t in transcripts:                                                          # verify.rs:909-913
            main.common_scalar(t.squeeze_challenge_scalar())
    lam = main.squeeze_challenge_scalar()                                              # :924
    agg = S.aggregate_fold([p for p, _c in proofs], lam)                               # :926-938
    left, right, _names = S.evaluate_multiopen_proof(ctx, S.OracleFieldChip(), pchip, agg)
    return left, right, plain, [c for _p, c in proofs], lam


def verify_aggregation_sharded_rank(pchip, circuits_local: List[CircuitProofs], global_index: List[int], n_total: int,
                                    allgather, ctx=None, make_transcript=None):
    """One rank of a SHARDED verify_aggregation_proofs_in_chip (first grain): this rank holds the proofs of
    `circuits_local`, whose positions in the aggregation order are global_index (circuits in order, proofs in order);
    n_total = N.  `allgather(payload: bytes) -> [bytes per rank]` is the only communication.  Restates what the reference
    does in one process (verify.rs:835-942) as the rank-local computation it distributes to:
      exchange 1   every proof's last squeeze, absorbed in aggregation order by the main transcript  (:909-913, :924)
      local fold   sum_j lambda^(N-1-g_j) * proof_j — the nested fold acc = acc * lambda + proof expanded   (:926-938)
      exchange 2   the partial (left, right) of every rank, added with the group law
    Returns (left, right, lambda), the same on every rank and the same as the one-process function."""
    ctx = ctx if ctx is not None else S.OracleCtx()
    mk = make_transcript or (lambda data: P.PoseidonTranscriptRead(data))
    proofs, squeezes = [], []
    for circuit in circuits_local:
        transcripts = []
        for i, (instances, data) in enumerate(circuit.proofs):
            t = mk(data)
            transcripts.append(t)
            _assigned, commitments = assign_instance_commitment(pchip, ctx, instances, circuit.cs, circuit.g_lagrange)
            p, _c, _vp = verify_single_proof_no_eval(t, pchip, ctx, commitments, circuit.cs, "%s_p%d" % (circuit.name, i))
            proofs.append(p)
        squeezes += [t.squeeze_challenge_scalar() for t in transcripts]
    assert len(proofs) == len(global_index)
    # exchange 1: {position, squeeze} records, n_total slots per rank
    rec = b"".join(g.to_bytes(4, "little") + O.fe_to_bytes(q) for g, q in zip(global_index, squeezes))
    rec += b"\xff" * (36 * (n_total - len(global_index)))
    slots = [None] * n_total
    for part in allgather(rec):
        for k in range(n_total):
            g = int.from_bytes(part[36 * k:36 * k + 4], "little")
            if g != 0xFFFFFFFF:
                assert slots[g] is None, "two ranks claim one proof position"
                slots[g] = int.from_bytes(part[36 * k + 4:36 * k + 36], "little")
    assert all(q is not None for q in slots), "a proof position nobody holds"
    main = mk(b"")
    for q in slots:
        main.common_scalar(q)
    lam = main.squeeze_challenge_scalar()
    # local fold with the powers the nested fold would have applied
    acc = None
    for p, g in zip(proofs, global_index):
        e = n_total - 1 - g
        term = p if e == 0 else S.MultiOpenProof(p.w_x * S.scalar(pow(lam, e, O.R)), p.w_g * S.scalar(pow(lam, e, O.R)))
        acc = term if acc is None else S.MultiOpenProof(acc.w_x + term.w_x, acc.w_g + term.w_g)
    if acc is None:
        left, right = None, None                                                       # the identity
    else:
        left, right, _names = S.evaluate_multiopen_proof(ctx, S.OracleFieldChip(), pchip, acc)
    # exchange 2
    mine = (O.aff_to_bytes(left) if left is not None else bytes(64)) + (O.aff_to_bytes(right) if right is not None else bytes(64))
    tot_l, tot_r = None, None
    for part in allgather(mine):
        tot_l = O.add(tot_l, O.aff_from_bytes(part[:64]))
        tot_r = O.add(tot_r, O.aff_from_bytes(part[64:]))
    return tot_l, tot_r, lam
