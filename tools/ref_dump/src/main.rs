//! Dumps what the REFERENCE computes, so that `oracle/` (and through it the HIP product) can be pinned against it.
//! Source-only here (no Rust toolchain in the build image); see Cargo.toml for the one command.
//!
//! For each of the reference's own test circuits (add_mul: `tests/systems/halo2/add_mul_test`, lookup:
//! `tests/systems/halo2/lookup_test`) it creates NPROOFS proofs exactly as the reference's test does
//! (add_mul_test/verify_aggregation.rs:62-103: ParamsKZG::setup, keygen, create_proof with PoseidonWrite and
//! Pcg32::seed_from_u64(0)), replays them through `verify_aggregation_proofs_in_chip` over MockFieldChip / MockEccChip /
//! PoseidonEncode (verify_aggregation.rs:105-149), and writes one JSON file per circuit holding every byte string the
//! restatement had to recall from unvendored crates:
//!
//!   params_bytes        ParamsKZG::write                       (layout of fs.py::read_params)
//!   g_lagrange          params.g_lagrange as 64-byte affine    (input of h2agg_bases_upload)
//!   s_g2, g2            128-byte affine G2 (x.c0 x.c1 y.c0 y.c1) (input of h2agg_final_pair_check)
//!   vk_blob             aggregate::serialize_vk(vk)            (the H2VK description h2agg_vk_create parses)
//!   vk_scalar           the value init_transcript absorbs      (verify.rs:57-70)
//!   proofs[i].transcript / .instances                          (PoseidonWrite bytes: compressed-point and to_repr layouts)
//!   proofs[i].challenges   every squeeze_challenge_scalar of the replay, in order (the sponge incl. State::default())
//!   lambda, w_x, w_g    the aggregation challenge and the final pair (verify.rs:909-941)
//!   advice_commitments  the fourth return value (verify.rs:852-856)
//!   pairing_ok          multi_miller_loop + final_exponentiation on the pair (verify.rs:733-739)
//!   final_pair_instances   final_pair_to_instances of the pair + plain instances (verify_circuit.rs:768-804)
//!   (one file per single-circuit aggregation and one, ref_multi_add_mul_lookup.json, with BOTH circuits in one aggregation)
//!   poseidon            State::default() words, the first three permutation outputs of the T = 9 spec on a fixed input,
//!                       mds[0][0..3], constants.start[0][0..3]  (the Grain generator and the absent MDS re-draw)
//!   encodings           G1Affine::to_bytes / to_repr samples for k*G, k = 1, 2, r-1, and the identity
//! and ref_chip_kats.json: MockEccChip::multi_exp / scalar_mul / add / sub over the inputs of the committed msm_kats.json and
//! point_kats.json (dump_chip_kats).
use std::{env, fs, marker::PhantomData, path::PathBuf};

use halo2_proofs::arithmetic::{CurveAffine, Field, FieldExt};
use halo2_proofs::plonk::{create_proof, keygen_pk, keygen_vk, Circuit, VerifyingKey};
use halo2_proofs::poly::commitment::{Params, ParamsProver};
use halo2_proofs::poly::kzg::commitment::{KZGCommitmentScheme, ParamsKZG, ParamsVerifierKZG};
use halo2_proofs::poly::kzg::multiopen::ProverGWC;
use halo2_proofs::transcript::{Challenge255, PoseidonWrite, TranscriptWriterBuffer};
use halo2_snark_aggregator_api::arith::{common::ArithCommonChip, ecc::ArithEccChip, field::ArithFieldChip};
use halo2_snark_aggregator_api::mock::arith::{ecc::MockEccChip, field::{MockChipCtx, MockFieldChip}};
use halo2_snark_aggregator_api::mock::transcript_encode::PoseidonEncode;
use halo2_snark_aggregator_api::systems::halo2::transcript::PoseidonTranscriptRead;
use halo2_snark_aggregator_api::systems::halo2::verify::{verify_aggregation_proofs_in_chip, CircuitProof, ProofData};
use halo2_snark_aggregator_api::tests::systems::halo2::add_mul_test::test_circuit::test_circuit_builder;
use halo2_snark_aggregator_api::tests::systems::halo2::lookup_test::test_circuit::test_circuit_builder as lookup_circuit_builder;
use halo2_snark_aggregator_api::transcript::read::TranscriptRead;
use halo2curves::bn256::{Bn256, Fr, G1Affine, G2Affine, G1};
use halo2curves::group::{ff::PrimeField, Curve, Group};
use halo2curves::pairing::{MillerLoopResult, MultiMillerLoop};
use rand::SeedableRng;
use rand_pcg::Pcg32;
use rand_xorshift::XorShiftRng;

const NPROOFS: usize = 2;

fn hex_fe(f: &Fr) -> String { hex::encode(f.to_repr()) }
fn hex_fq(f: &halo2curves::bn256::Fq) -> String { hex::encode(f.to_repr()) }
fn aff64(p: &G1Affine) -> String {
    // the C ABI's affine encoding: x || y canonical little-endian, identity = 64 zero bytes
    let c = p.coordinates();
    if bool::from(c.is_some()) { let c = c.unwrap(); format!("{}{}", hex_fq(c.x()), hex_fq(c.y())) } else { "00".repeat(64) }
}
fn g2_128(p: &G2Affine) -> String {
    let c = p.coordinates().unwrap();
    format!("{}{}{}{}", hex_fq(&c.x().c0), hex_fq(&c.x().c1), hex_fq(&c.y().c0), hex_fq(&c.y().c1))
}
fn json_list(items: &[String]) -> String { format!("[{}]", items.iter().map(|s| format!("\"{}\"", s)).collect::<Vec<_>>().join(", ")) }

/// A TranscriptRead that records every squeezed challenge of the replay.
struct Recorder<'a, T> { inner: T, log: &'a std::cell::RefCell<Vec<Fr>> }
impl<'a, T: TranscriptRead<MockEccChip<G1Affine, halo2_proofs::plonk::Error>>> TranscriptRead<MockEccChip<G1Affine, halo2_proofs::plonk::Error>> for Recorder<'a, T> {
    fn read_point(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, p: &MockEccChip<G1Affine, halo2_proofs::plonk::Error>) -> Result<G1, halo2_proofs::plonk::Error> { self.inner.read_point(c, n, s, p) }
    fn read_constant_point(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, p: &MockEccChip<G1Affine, halo2_proofs::plonk::Error>) -> Result<G1, halo2_proofs::plonk::Error> { self.inner.read_constant_point(c, n, s, p) }
    fn read_scalar(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>) -> Result<Fr, halo2_proofs::plonk::Error> { self.inner.read_scalar(c, n, s) }
    fn read_constant_scalar(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>) -> Result<Fr, halo2_proofs::plonk::Error> { self.inner.read_constant_scalar(c, n, s) }
    fn squeeze_challenge_scalar(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>) -> Result<Fr, halo2_proofs::plonk::Error> {
        let v = self.inner.squeeze_challenge_scalar(c, n, s)?;
        self.log.borrow_mut().push(v);
        Ok(v)
    }
    fn common_point(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, p: &MockEccChip<G1Affine, halo2_proofs::plonk::Error>, pt: &G1) -> Result<(), halo2_proofs::plonk::Error> { self.inner.common_point(c, n, s, p, pt) }
    fn common_scalar(&mut self, c: &mut MockChipCtx, n: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, s: &MockFieldChip<Fr, halo2_proofs::plonk::Error>, v: &Fr) -> Result<(), halo2_proofs::plonk::Error> { self.inner.common_scalar(c, n, s, v) }
}

/// One circuit's proofs, made exactly as the reference's test makes them (add_mul_test/verify_aggregation.rs:62-103).
struct Made {
    name: String,
    vk: VerifyingKey<G1Affine>,
    proofs: Vec<Vec<u8>>,
    instances: Vec<Vec<Vec<Vec<Fr>>>>,
}
fn make<C: Circuit<Fr> + Clone>(name: &str, params: &ParamsKZG<Bn256>, circuits: Vec<(C, Vec<Vec<Vec<Fr>>>)>, template: C) -> Made {
    let vk: VerifyingKey<G1Affine> = keygen_vk(params, &template).expect("keygen_vk");
    let mut proofs = vec![];
    let mut all_instances = vec![];
    for (circuit, instances) in circuits.iter() {
        let pk = keygen_pk(params, keygen_vk(params, &template).unwrap(), circuit).expect("keygen_pk");
        let mut transcript = PoseidonWrite::<Vec<u8>, G1Affine, Challenge255<G1Affine>>::init(vec![]);
        let i1: Vec<Vec<&[Fr]>> = instances.iter().map(|x| x.iter().map(|y| &y[..]).collect()).collect();
        let i2: Vec<&[&[Fr]]> = i1.iter().map(|x| &x[..]).collect();
        create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC<Bn256>, _, _, _, _>(params, &pk, &[circuit.clone()], &i2[..], Pcg32::seed_from_u64(0), &mut transcript).expect("create_proof");
        proofs.push(transcript.finalize());
        all_instances.push(instances.clone());
    }
    Made { name: String::from(name), vk, proofs, instances: all_instances }
}

/// The reference's replay over the Mock chips (verify_aggregation.rs:105-149) of ONE aggregation holding every circuit of
/// `made` (several circuits = the multi-circuit case of verify.rs:859-915: each circuit's squeezes enter the aggregation
/// transcript after its proofs), dumped as ref_<file>.json:
///   shared: k, params_bytes, g_lagrange, s_g2, g2, lambda, w_x, w_g, pairing_ok, advice_commitments (aggregation order),
///           final_pair_instances = final_pair_to_instances(w_x, w_g, plain instances)   (verify_circuit.rs:768-804)
///   circuits[]: circuit, vk_blob, vk_scalar, proofs[] {transcript, instances, challenges}
fn replay_and_dump(file: &str, k: u32, params: &ParamsKZG<Bn256>, made: &[Made], out_dir: &PathBuf) {
    let params_verifier: &ParamsVerifierKZG<Bn256> = params.verifier_params();
    let nchip = MockFieldChip::<Fr, halo2_proofs::plonk::Error>::default();
    let schip = MockFieldChip::<Fr, halo2_proofs::plonk::Error>::default();
    let pchip = MockEccChip::<G1Affine, halo2_proofs::plonk::Error>::default();
    let ctx = &mut MockChipCtx::default();
    let total: usize = made.iter().map(|m| m.proofs.len()).sum();
    let logs: Vec<std::cell::RefCell<Vec<Fr>>> = (0..total + 1).map(|_| std::cell::RefCell::new(vec![])).collect();
    let mut circuit_proofs = vec![];
    let mut g = 0usize;
    for m in made.iter() {
        let mut list = vec![];
        for i in 0..m.proofs.len() {
            let t = PoseidonTranscriptRead::<_, G1Affine, _, PoseidonEncode, 9usize, 8usize>::new(&m.proofs[i][..], ctx, &nchip, 8usize, 63usize).unwrap();
            list.push(ProofData { instances: &m.instances[i], transcript: Recorder { inner: t, log: &logs[g] }, key: format!("{}_p{}", m.name, i), _phantom: PhantomData });
            g += 1;
        }
        circuit_proofs.push(CircuitProof { name: m.name.clone(), vk: &m.vk, params: params_verifier, proofs: list });
    }
    let empty: Vec<u8> = vec![];
    let mut main_t = Recorder { inner: PoseidonTranscriptRead::<_, G1Affine, _, PoseidonEncode, 9usize, 8usize>::new(&empty[..], ctx, &nchip, 8usize, 63usize).unwrap(), log: &logs[total] };
    let (w_x, w_g, plain, commits) = verify_aggregation_proofs_in_chip(ctx, &nchip, &schip, &pchip, circuit_proofs, &mut main_t).unwrap();
    let (wx, wg) = (w_x.to_affine(), w_g.to_affine());
    // verify.rs:733-739
    let s_g2_prepared = <Bn256 as MultiMillerLoop>::G2Prepared::from(params_verifier.s_g2());
    let n_g2_prepared = <Bn256 as MultiMillerLoop>::G2Prepared::from(-params_verifier.g2());
    let ok = bool::from(Bn256::multi_miller_loop(&[(&wx, &s_g2_prepared), (&wg, &n_g2_prepared)]).final_exponentiation().is_identity());
    // the verify circuit's public inputs of this pair (halo2-snark-aggregator-circuit/src/verify_circuit.rs:768-804)
    let fp_instances = halo2_snark_aggregator_circuit::verify_circuit::final_pair_to_instances::<G1Affine, Bn256>(&(wx, wg, plain.clone()));
    let mut params_bytes = vec![];
    params.write(&mut params_bytes).unwrap();
    let mut o = String::from("{\n");
    o += &format!("  \"k\": {}, \"nproofs\": {},\n", k, total);
    o += &format!("  \"params_bytes\": \"{}\",\n", hex::encode(&params_bytes));
    o += &format!("  \"g_lagrange\": \"{}\",\n", params.g_lagrange().iter().map(aff64).collect::<String>());   // (Params::g_lagrange accessor of the pinned halo2_proofs)
    o += &format!("  \"s_g2\": \"{}\", \"g2\": \"{}\",\n", g2_128(&params_verifier.s_g2()), g2_128(&params_verifier.g2()));
    o += "  \"circuits\": [\n";
    let mut g = 0usize;
    for (ci, m) in made.iter().enumerate() {
        // the value init_transcript absorbs (verify.rs:57-70): blake2b("Halo2-Verify-Key") over the pinned vk, from_bytes_wide
        let vk_scalar = {
            let mut h = blake2b_simd::Params::new().hash_length(64).personal(b"Halo2-Verify-Key").to_state();
            h.update(format!("{:?}", m.vk.pinned()).as_bytes());
            Fr::from_bytes_wide(h.finalize().as_array())
        };
        let vk_blob = h2agg_sys::aggregate::serialize_vk(&m.vk, k, vk_scalar);
        o += &format!("    {{\"circuit\": \"{}\", \"vk_blob\": \"{}\", \"vk_scalar\": \"{}\", \"proofs\": [\n", m.name, hex::encode(&vk_blob), hex_fe(&vk_scalar));
        for i in 0..m.proofs.len() {
            let cols: Vec<String> = m.instances[i][0].iter().map(|col| col.iter().map(hex_fe).collect::<String>()).collect();
            o += &format!("      {{\"transcript\": \"{}\", \"instances\": {}, \"challenges\": {}}}{}\n", hex::encode(&m.proofs[i]), json_list(&cols),
                json_list(&logs[g].borrow().iter().map(hex_fe).collect::<Vec<_>>()), if i + 1 < m.proofs.len() { "," } else { "" });
            g += 1;
        }
        o += &format!("    ]}}{}\n", if ci + 1 < made.len() { "," } else { "" });
    }
    o += "  ],\n";
    o += &format!("  \"lambda\": \"{}\",\n", hex_fe(logs[total].borrow().last().unwrap()));
    o += &format!("  \"w_x\": \"{}\", \"w_g\": \"{}\", \"pairing_ok\": {},\n", aff64(&wx), aff64(&wg), ok);
    o += &format!("  \"final_pair_instances\": {},\n", json_list(&fp_instances.iter().map(hex_fe).collect::<Vec<_>>()));
    o += &format!("  \"advice_commitments\": [{}]\n", commits.iter().map(|per| json_list(&per.iter().map(|p| aff64(&p.to_affine())).collect::<Vec<_>>())).collect::<Vec<_>>().join(", "));
    o += "}\n";
    fs::write(out_dir.join(format!("ref_{}.json", file)), o).unwrap();
}

fn dump_primitives(out_dir: &PathBuf) {
    // Poseidon: State::default(), the spec's first constants, three permutations of a fixed absorb pattern (T = 9, RATE = 8)
    let spec = poseidon::Spec::<Fr, 9, 8>::new(8, 63);
    let mut sponge = poseidon::Poseidon::<Fr, 9, 8>::new(8, 63);
    let mut outs = vec![];
    sponge.update(&[Fr::from(1), Fr::from(2), Fr::from(3)]);
    outs.push(hex_fe(&sponge.squeeze()));
    sponge.update(&(0..8u64).map(Fr::from).collect::<Vec<_>>());
    outs.push(hex_fe(&sponge.squeeze()));
    outs.push(hex_fe(&sponge.squeeze()));
    let state0 = poseidon::State::<Fr, 9>::default();
    let mds = spec.mds_matrices().mds().rows();
    let start = spec.constants().start();
    // encodings
    let g = G1::generator();
    let samples: Vec<(String, G1Affine)> = vec![("1".into(), g.to_affine()), ("2".into(), (g + g).to_affine()), ("r-1".into(), (g * (-Fr::one())).to_affine()), ("identity".into(), G1Affine::identity())];
    let mut o = String::from("{\n");
    o += &format!("  \"poseidon_state_default\": {},\n", json_list(&state0.words().iter().map(hex_fe).collect::<Vec<_>>()));
    o += &format!("  \"poseidon_squeezes\": {},\n", json_list(&outs));
    o += &format!("  \"poseidon_mds_row0\": {},\n", json_list(&mds[0].iter().map(hex_fe).collect::<Vec<_>>()));
    o += &format!("  \"poseidon_start0\": {},\n", json_list(&start[0].iter().map(hex_fe).collect::<Vec<_>>()));
    o += "  \"g1_encodings\": [\n";
    for (i, (k, p)) in samples.iter().enumerate() {
        use halo2curves::group::GroupEncoding;
        o += &format!("    {{\"k\": \"{}\", \"affine\": \"{}\", \"compressed\": \"{}\"}}{}\n", k, aff64(p), hex::encode(p.to_bytes()), if i + 1 < samples.len() { "," } else { "" });
    }
    o += "  ],\n";
    o += &format!("  \"fr_to_repr_of_5\": \"{}\", \"fr_root_of_unity\": \"{}\", \"fr_delta\": \"{}\"\n", hex_fe(&Fr::from(5)), hex_fe(&Fr::root_of_unity()), hex_fe(&Fr::DELTA));
    o += "}\n";
    fs::write(out_dir.join("ref_primitives.json"), o).unwrap();
}

/// every `"key": "<hex>"` of a JSON text, in order (the committed fixtures are flat lists of objects with hex-string fields;
/// serde is not among the workspace's dependencies)
fn hex_fields(text: &str, key: &str) -> Vec<Vec<u8>> {
    let pat = format!("\"{}\": \"", key);
    let mut out = vec![];
    let mut at = 0;
    while let Some(i) = text[at..].find(&pat) {
        let start = at + i + pat.len();
        let end = start + text[start..].find('"').unwrap();
        out.push(hex::decode(&text[start..end]).unwrap());
        at = end;
    }
    out
}
fn fq_from(b: &[u8]) -> halo2curves::bn256::Fq {
    let mut r = [0u8; 32];
    r.copy_from_slice(b);
    Option::from(halo2curves::bn256::Fq::from_repr(r)).expect("canonical coordinate")
}
fn fr_from(b: &[u8]) -> Fr {
    let mut r = [0u8; 32];
    r.copy_from_slice(b);
    Option::from(Fr::from_repr(r)).expect("canonical scalar")
}
/// the C ABI's affine encoding back into a point (64 zero bytes: the identity)
fn g1_from_aff64(b: &[u8]) -> G1 {
    if b.iter().all(|&x| x == 0) { return G1::identity(); }
    let p: G1Affine = Option::from(G1Affine::from_xy(fq_from(&b[..32]), fq_from(&b[32..64]))).expect("point on the curve");
    p.into()
}

/// Rows a1-a8 of SURVEY.md section 8 pinned in the same run as the pipeline: the reference's OWN MockEccChip (multi_exp =
/// mock/arith/ecc.rs:106-129, scalar_mul :48-62, add / sub :30-46) over the INPUTS of the committed, oracle-made fixtures
/// tests/golden/msm_kats.json and point_kats.json; the outputs go to ref_chip_kats.json, which tests/test_ref_golden.py holds
/// against those fixtures' own outputs (the ones the oracle and the HIP kernels already reproduce).
fn dump_chip_kats(golden: &PathBuf) {
    type E = halo2_proofs::plonk::Error;
    let pchip = MockEccChip::<G1Affine, E>::default();
    let mut ctx = MockChipCtx::default();
    let msm = fs::read_to_string(golden.join("msm_kats.json")).expect("tests/golden/msm_kats.json");
    let (bases, scalars) = (hex_fields(&msm, "bases_aff"), hex_fields(&msm, "scalars"));
    let mut msm_out = vec![];
    for (b, sc) in bases.iter().zip(scalars.iter()) {
        let pts: Vec<G1> = b.chunks(64).map(g1_from_aff64).collect();
        let ss: Vec<Fr> = sc.chunks(32).map(fr_from).collect();
        let r = pchip.multi_exp(&mut ctx, pts, ss).unwrap();
        msm_out.push(aff64(&r.to_affine()));
    }
    let pk = fs::read_to_string(golden.join("point_kats.json")).expect("tests/golden/point_kats.json");
    let mut mul_out = vec![];
    for (b, sc) in hex_fields(&pk, "base_aff").iter().zip(hex_fields(&pk, "scalar").iter()) {
        let r = pchip.scalar_mul(&mut ctx, &fr_from(sc), &g1_from_aff64(b)).unwrap();
        mul_out.push(aff64(&r.to_affine()));
    }
    // (the add fixtures hold Jacobian triples x || y || z, 96 bytes: the reference's CurveExt has no such constructor in
    // halo2curves 0.2.1, so they are rebuilt from their affine form x / z^2, y / z^3)
    let jac = |b: &[u8]| -> G1 {
        let (x, y, z) = (fq_from(&b[..32]), fq_from(&b[32..64]), fq_from(&b[64..96]));
        if bool::from(z.is_zero()) { return G1::identity(); }
        let zi = z.invert().unwrap();
        let zi2 = zi.square();
        let p: G1Affine = Option::from(G1Affine::from_xy(x * zi2, y * zi2 * zi)).expect("point on the curve");
        p.into()
    };
    let (mut sum_out, mut diff_out) = (vec![], vec![]);
    for (a, b) in hex_fields(&pk, "a_jac").iter().zip(hex_fields(&pk, "b_jac").iter()) {
        let (pa, pb) = (jac(a), jac(b));
        sum_out.push(aff64(&pchip.add(&mut ctx, &pa, &pb).unwrap().to_affine()));
        diff_out.push(aff64(&pchip.sub(&mut ctx, &pa, &pb).unwrap().to_affine()));
    }
    let o = format!("{{\n  \"multi_exp_out_aff\": {},\n  \"scalar_mul_out_aff\": {},\n  \"add_sum_aff\": {},\n  \"sub_diff_aff\": {}\n}}\n",
                    json_list(&msm_out), json_list(&mul_out), json_list(&sum_out), json_list(&diff_out));
    fs::write(golden.join("ref_chip_kats.json"), o).unwrap();
}

fn main() {
    let out_dir = PathBuf::from(env::args().nth(1).expect("usage: ref_dump <tests/golden directory>"));
    fs::create_dir_all(&out_dir).unwrap();
    dump_primitives(&out_dir);
    dump_chip_kats(&out_dir);
    // add_mul: c = 7 a^2 b^2 (verify_aggregation.rs:75-82), fixed witnesses instead of the clock-seeded ones
    let add_mul = |rng: &mut XorShiftRng| -> Vec<_> {
        (0..NPROOFS).map(|_| {
            let (a, b) = (Fr::random(&mut *rng), Fr::random(&mut *rng));
            (test_circuit_builder(a, b), vec![vec![vec![Fr::from(7) * a.square() * b.square()]]])
        }).collect()
    };
    // lookup: the reference's lookup test circuit with its own instance column (lookup_test/verify_aggregation.rs:50-70)
    let odd_lookup = vec![Fr::from(1), Fr::from(3), Fr::from(5), Fr::from(7), Fr::from(9)];
    let lookup = || -> Vec<_> { (0..NPROOFS).map(|_| (lookup_circuit_builder(), vec![vec![odd_lookup.clone()]])).collect() };
    let mut rng = XorShiftRng::seed_from_u64(0xADD);
    {   // one circuit per aggregation, at the sizes the reference's own tests use (K = 10 / K = 6)
        let p10 = ParamsKZG::<Bn256>::setup(10, &mut XorShiftRng::seed_from_u64(0x4832_4147));     // fixed: the dump is reproducible
        replay_and_dump("test_circuit_add_mul", 10, &p10, &[make("test_circuit_add_mul", &p10, add_mul(&mut rng), test_circuit_builder(Fr::zero(), Fr::zero()))], &out_dir);
        let p6 = ParamsKZG::<Bn256>::setup(6, &mut XorShiftRng::seed_from_u64(0x4832_4147));
        replay_and_dump("test_circuit_lookup", 6, &p6, &[make("test_circuit_lookup", &p6, lookup(), lookup_circuit_builder())], &out_dir);
    }
    {   // BOTH circuits in ONE aggregation over one SRS (the multi-circuit fold of verify.rs:859-915, :924-938; lookups +
        // several keys + the squeeze order across circuits), K = 10 for both
        let p = ParamsKZG::<Bn256>::setup(10, &mut XorShiftRng::seed_from_u64(0x4832_4148));
        let made = [make("test_circuit_add_mul", &p, add_mul(&mut rng), test_circuit_builder(Fr::zero(), Fr::zero())),
                    make("test_circuit_lookup", &p, lookup(), lookup_circuit_builder())];
        replay_and_dump("multi_add_mul_lookup", 10, &p, &made, &out_dir);
    }
    eprintln!("wrote {}/ref_*.json — now run: python -m pytest tests/test_ref_golden.py", out_dir.display());
}
