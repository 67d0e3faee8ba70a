//! GPU-backed drop-in for `halo2_snark_aggregator_api::mock::arith::ecc::MockEccChip`.
//!
//! SOURCE ONLY — never compiled in the build image (no Rust toolchain there).  Layout facts about
//! halo2curves 0.2.1 used below (`to_repr()` = 32-byte little-endian canonical integer,
//! `G1Affine { x, y }`, identity = (0, 0)) must be re-verified where Rust exists.
//!
//! Only `multi_exp` (and, optionally, batched `scalar_mul_constant` for the instance commitment)
//! crosses the FFI; every other method stays exactly what the reference's Mock chip does, so
//! `AssignedPoint = C::CurveExt` is preserved and `-api / -circuit / -sdk` compile unchanged.
use group::{Curve, Group};
use halo2_proofs::arithmetic::{CurveAffine, FieldExt};
use halo2_snark_aggregator_api::arith::{common::ArithCommonChip, ecc::ArithEccChip};
use halo2_snark_aggregator_api::mock::arith::ecc::MockEccChip;
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct h2agg_ctx {
    _private: [u8; 0],
}

extern "C" {
    // include/h2agg.h
    fn h2agg_create(device_ordinal: c_int, out: *mut *mut h2agg_ctx) -> c_int;
    fn h2agg_destroy(ctx: *mut h2agg_ctx);
    fn h2agg_last_error(ctx: *const h2agg_ctx) -> *const c_char;
    fn h2agg_g1_msm(ctx: *mut h2agg_ctx, bases_aff: *const u8, scalars: *const u8, n: usize, out_jac: *mut u8) -> c_int;
    fn h2agg_g1_msm_jac(ctx: *mut h2agg_ctx, points_jac: *const u8, scalars: *const u8, n: usize, out_jac: *mut u8) -> c_int;
    fn h2agg_g1_batch_scalar_mul(ctx: *mut h2agg_ctx, bases_aff: *const u8, scalars: *const u8, n: usize, out_jac: *mut u8) -> c_int;
    // page-locked marshalling buffers: for large multi_exps, serialise points / scalars straight into memory from
    // h2agg_host_alloc instead of a Vec<u8> (the 96 B/point then cross PCIe at link rate, sliced under the compute)
    #[allow(dead_code)]
    fn h2agg_host_alloc(ctx: *mut h2agg_ctx, bytes: usize, out: *mut *mut u8) -> c_int;
    #[allow(dead_code)]
    fn h2agg_host_free(ctx: *mut h2agg_ctx, p: *mut u8) -> c_int;
}

/// Wraps the reference's own Mock chip and forwards everything to it except `multi_exp`.
pub struct GpuEccChip<C: CurveAffine, E> {
    host: MockEccChip<C, E>,
    gpu: *mut h2agg_ctx, // single-thread-affine, like the reference's use of the chips
}

impl<C: CurveAffine, E> Default for GpuEccChip<C, E> {
    fn default() -> Self {
        let mut gpu = std::ptr::null_mut();
        let rc = unsafe { h2agg_create(0, &mut gpu) };
        assert_eq!(rc, 0, "h2agg_create failed: a HIP device is required");
        Self { host: MockEccChip::default(), gpu }
    }
}
impl<C: CurveAffine, E> Drop for GpuEccChip<C, E> {
    fn drop(&mut self) {
        unsafe { h2agg_destroy(self.gpu) }
    }
}

fn put_fe<F: FieldExt>(dst: &mut Vec<u8>, f: &F) {
    dst.extend_from_slice(f.to_repr().as_ref()); // 32-byte LE canonical
}
fn affine_bytes<C: CurveAffine>(dst: &mut Vec<u8>, p: &C) {
    match Option::<_>::from(p.coordinates()) {
        Some(c) => {
            let c: halo2_proofs::arithmetic::Coordinates<C> = c;
            put_fe(dst, c.x());
            put_fe(dst, c.y());
        }
        None => dst.extend_from_slice(&[0u8; 64]), // identity = 64 zero bytes
    }
}
fn fe_from<F: FieldExt>(b: &[u8]) -> F {
    let mut repr = F::Repr::default();
    repr.as_mut().copy_from_slice(b);
    Option::from(F::from_repr(repr)).expect("canonical field element")
}
/// Jacobian (x, y, z) from the library -> C::CurveExt, via the affine point (one Fq inversion on the host,
/// the same `to_affine` the reference performs at verify_circuit.rs:180,200).
fn curve_from_jac<C: CurveAffine>(b: &[u8; 96]) -> C::CurveExt {
    let (x, y, z): (C::Base, C::Base, C::Base) = (fe_from(&b[0..32]), fe_from(&b[32..64]), fe_from(&b[64..96]));
    if bool::from(z.is_zero()) {
        return C::CurveExt::identity();
    }
    let zi = z.invert().unwrap();
    let zi2 = zi.square();
    C::from_xy(x * zi2, y * zi2 * zi).unwrap().to_curve()
}

type Host<C, E> = MockEccChip<C, E>;
type Ctx<C, E> = <Host<C, E> as ArithCommonChip>::Context;
type Pt<C, E> = <Host<C, E> as ArithCommonChip>::AssignedValue;
type Sc<C, E> = <Host<C, E> as ArithEccChip>::AssignedScalar;

impl<C: CurveAffine, E> ArithCommonChip for GpuEccChip<C, E> {
    type Context = Ctx<C, E>;
    type Value = <Host<C, E> as ArithCommonChip>::Value;
    type AssignedValue = Pt<C, E>;
    type Error = E;
    // element-wise group operations: forwarded to the host chip unchanged
    fn add(&self, ctx: &mut Self::Context, a: &Pt<C, E>, b: &Pt<C, E>) -> Result<Pt<C, E>, E> {
        self.host.add(ctx, a, b)
    }
    fn sub(&self, ctx: &mut Self::Context, a: &Pt<C, E>, b: &Pt<C, E>) -> Result<Pt<C, E>, E> {
        self.host.sub(ctx, a, b)
    }
    fn assign_zero(&self, ctx: &mut Self::Context) -> Result<Pt<C, E>, E> {
        self.host.assign_zero(ctx)
    }
    fn assign_one(&self, ctx: &mut Self::Context) -> Result<Pt<C, E>, E> {
        self.host.assign_one(ctx)
    }
    fn assign_const(&self, ctx: &mut Self::Context, c: C) -> Result<Pt<C, E>, E> {
        self.host.assign_const(ctx, c)
    }
    fn assign_var(&self, ctx: &mut Self::Context, v: C) -> Result<Pt<C, E>, E> {
        self.host.assign_var(ctx, v)
    }
    fn to_value(&self, v: &Pt<C, E>) -> Result<C, E> {
        self.host.to_value(v)
    }
    fn normalize(&self, ctx: &mut Self::Context, v: &Pt<C, E>) -> Result<Pt<C, E>, E> {
        self.host.normalize(ctx, v)
    }
}

impl<C: CurveAffine, E> ArithEccChip for GpuEccChip<C, E> {
    type Point = C;
    type AssignedPoint = Pt<C, E>;
    type Scalar = <Host<C, E> as ArithEccChip>::Scalar;
    type AssignedScalar = Sc<C, E>;
    type Native = <Host<C, E> as ArithEccChip>::Native;
    type AssignedNative = <Host<C, E> as ArithEccChip>::AssignedNative;
    type ScalarChip = <Host<C, E> as ArithEccChip>::ScalarChip;
    type NativeChip = <Host<C, E> as ArithEccChip>::NativeChip;

    // single products stay on the host chip (a kernel launch costs more than one scalar multiplication)
    fn scalar_mul(&self, ctx: &mut Self::Context, lhs: &Sc<C, E>, rhs: &Pt<C, E>) -> Result<Pt<C, E>, E> {
        self.host.scalar_mul(ctx, lhs, rhs)
    }
    fn scalar_mul_constant(&self, ctx: &mut Self::Context, lhs: &Sc<C, E>, rhs: C) -> Result<Pt<C, E>, E> {
        self.host.scalar_mul_constant(ctx, lhs, rhs)
    }

    /// mock/arith/ecc.rs:106-129 with the loop of scalar muls replaced by one GPU MSM.
    fn multi_exp(&self, ctx: &mut Self::Context, points: Vec<Pt<C, E>>, scalars: Vec<Sc<C, E>>) -> Result<Pt<C, E>, E> {
        // observable side effect kept: `Display for MockChipCtx` prints point_list.len()
        ctx.point_list = points.iter().map(|x| format!("{:?}", x)).collect();
        let n = points.len().min(scalars.len());
        // The points go over as they are — projective x || y || z — and are normalised on the device (h2agg_g1_msm_jac):
        // `batch_normalize` over 2^20 points is a third of a second on one host core, the whole MSM is 2 ms on the GPU.
        // (`jacobian_coordinates()` is halo2curves' accessor for the three coordinates of a `CurveExt`.)
        let (mut pb, mut sb) = (Vec::with_capacity(96 * n), Vec::with_capacity(32 * n));
        for i in 0..n {
            let (x, y, z) = points[i].jacobian_coordinates();
            put_fe(&mut pb, &x);
            put_fe(&mut pb, &y);
            put_fe(&mut pb, &z);
            put_fe(&mut sb, &scalars[i]);
        }
        let mut out = [0u8; 96];
        let rc = unsafe { h2agg_g1_msm_jac(self.gpu, pb.as_ptr(), sb.as_ptr(), n, out.as_mut_ptr()) };
        if rc != 0 {
            // H2AGG_ERR_EMPTY (3) reproduces the reference's `acc.unwrap()` panic on zero pairs
            let msg = unsafe { std::ffi::CStr::from_ptr(h2agg_last_error(self.gpu)) };
            panic!("h2agg_g1_msm failed ({}): {:?}", rc, msg);
        }
        Ok(curve_from_jac::<C>(&out))
    }
}

// =====================================================================================================================
// Round 2: the entry points ABOVE multi_exp — what `calc_verify_circuit_final_pair`
// (halo2-snark-aggregator-circuit/src/verify_circuit.rs:114-201) calls — offloaded as a whole.
// SOURCE ONLY like the rest of this file; the accessors on `VerifyingKey` / `ConstraintSystem` are the ones the reference
// itself uses (file:line cited per field), halo2_proofs being unvendored they could not be compiled against here.
// =====================================================================================================================
pub mod aggregate {
    use super::*;
    use halo2_proofs::plonk::{Any, Expression, VerifyingKey};
    use halo2_proofs::poly::kzg::commitment::ParamsKZG;
    use halo2curves::pairing::MultiMillerLoop;

    #[repr(C)]
    pub struct h2agg_vk {
        _private: [u8; 0],
    }
    #[repr(C)]
    pub struct h2agg_circuit_proofs {
        pub vk: *const h2agg_vk,
        pub name: *const c_char,
        pub g_lagrange: u64,
        pub nproofs: usize,
        pub transcripts: *const *const u8,
        pub transcript_lens: *const usize,
        pub instances: *const *const u8,
        pub instance_lens: *const u32,
    }
    extern "C" {
        fn h2agg_bases_upload(ctx: *mut h2agg_ctx, bases_aff: *const u8, n: usize, handle_out: *mut u64) -> c_int;
        fn h2agg_bases_precompute(ctx: *mut h2agg_ctx, handle: u64, window_bits: c_int) -> c_int;
        fn h2agg_instance_commitment(ctx: *mut h2agg_ctx, g_lagrange: u64, instance: *const u8, len: usize, max_len: usize, out_jac: *mut u8) -> c_int;
        fn h2agg_vk_create(ctx: *mut h2agg_ctx, blob: *const u8, len: usize, out: *mut *mut h2agg_vk) -> c_int;
        fn h2agg_vk_destroy(vk: *mut h2agg_vk);
        fn h2agg_verify_aggregation(
            ctx: *mut h2agg_ctx, circuits: *const h2agg_circuit_proofs, ncircuits: usize, s_g2: *const u8, g2: *const u8,
            left_aff: *mut u8, right_aff: *mut u8, lambda_out: *mut u8, pairing_ok: *mut c_int,
        ) -> c_int;
        // one process per GPU: rank 0 makes the id, every rank joins, then one exchange per aggregation
        pub fn h2agg_comm_unique_id(out: *mut u8) -> c_int;
        pub fn h2agg_comm_init_rank(ctx: *mut h2agg_ctx, id: *const u8, rank: c_int, nranks: c_int) -> c_int;
        pub fn h2agg_allgather_add_points(ctxs: *mut *mut h2agg_ctx, nctx: c_int, partial_jac: *const u8, npts: usize, out_aff: *mut u8) -> c_int;
    }

    /// `params.g_lagrange` resident on the device, with fixed-base levels (once per SRS): the bases of every
    /// assign_instance_commitment (verify.rs:623-635).
    pub fn upload_g_lagrange<E: MultiMillerLoop>(gpu: *mut h2agg_ctx, params: &ParamsKZG<E>) -> u64 {
        let mut buf = Vec::with_capacity(64 * params.g_lagrange.len());
        for p in params.g_lagrange.iter() {
            affine_bytes(&mut buf, p);
        }
        let mut h = 0u64;
        assert_eq!(unsafe { h2agg_bases_upload(gpu, buf.as_ptr(), params.g_lagrange.len(), &mut h) }, 0);
        let _ = unsafe { h2agg_bases_precompute(gpu, h, 0) }; // refused for tables above ~2^18 points: the plain MSM is used
        h
    }

    /// One column of assign_instance_commitment (verify.rs:596-640) as one MSM.
    pub fn instance_commitment<C: CurveAffine>(gpu: *mut h2agg_ctx, g_lagrange: u64, instance: &[C::ScalarExt], n: usize, blinding_factors: usize) -> C::CurveExt {
        let mut sb = Vec::with_capacity(32 * instance.len());
        for s in instance {
            put_fe(&mut sb, s);
        }
        let mut out = [0u8; 96];
        let rc = unsafe { h2agg_instance_commitment(gpu, g_lagrange, sb.as_ptr(), instance.len(), n - (blinding_factors + 1), out.as_mut_ptr()) };
        assert_eq!(rc, 0, "assert!(instance.len() <= params.n() - (blinding_factors + 1)) or device failure");
        curve_from_jac::<C>(&out)
    }

    fn put_expr<F: FieldExt>(out: &mut Vec<u8>, e: &Expression<F>) {
        // postfix bytecode of include/h2agg.h; the variants are the ones convert_expression walks (verify.rs:175-204)
        match e {
            Expression::Constant(c) => { out.push(0); put_fe(out, c); }
            Expression::Selector(_) => panic!("virtual selectors are removed during optimization"), // expression.rs:33-35
            Expression::Fixed(q) => { out.push(1); out.extend_from_slice(&(q.index() as u32).to_le_bytes()); }
            Expression::Advice(q) => { out.push(2); out.extend_from_slice(&(q.index() as u32).to_le_bytes()); }
            Expression::Instance(q) => { out.push(3); out.extend_from_slice(&(q.index() as u32).to_le_bytes()); }
            Expression::Challenge(ch) => { out.push(4); out.extend_from_slice(&(ch.index() as u32).to_le_bytes()); }
            Expression::Negated(a) => { put_expr(out, a); out.push(5); }
            Expression::Sum(a, b) => { put_expr(out, a); put_expr(out, b); out.push(6); }
            Expression::Product(a, b) => { put_expr(out, a); put_expr(out, b); out.push(7); }
            Expression::Scaled(a, f) => { put_expr(out, a); out.push(8); put_fe(out, f); }
        }
    }
    fn put_exprs<F: FieldExt>(out: &mut Vec<u8>, es: &[Expression<F>]) {
        out.extend_from_slice(&(es.len() as u32).to_le_bytes());
        for e in es {
            let mut code = Vec::new();
            put_expr(&mut code, e);
            out.extend_from_slice(&(code.len() as u32).to_le_bytes());
            out.extend_from_slice(&code);
            out.resize(out.len() + (4 - code.len() % 4) % 4, 0);
        }
    }

    /// The "H2VK" description of include/h2agg.h from a halo2 verifying key: every field with the accessor the reference
    /// reads it through.  `vk_scalar` = the value init_transcript absorbs (verify.rs:57-70).
    pub fn serialize_vk<C: CurveAffine>(vk: &VerifyingKey<C>, k: u32, vk_scalar: C::ScalarExt) -> Vec<u8> {
        let cs = vk.cs();
        let mut o = Vec::new();
        let u32le = |o: &mut Vec<u8>, v: usize| o.extend_from_slice(&(v as u32).to_le_bytes());
        u32le(&mut o, 0x4B56_3248);
        u32le(&mut o, 1);
        u32le(&mut o, k as usize);
        u32le(&mut o, cs.num_advice_columns());                 // verify.rs:352
        u32le(&mut o, cs.num_instance_columns);                 // verify.rs:592
        u32le(&mut o, cs.num_challenges());                     // verify.rs:359
        u32le(&mut o, cs.degree());                             // verify.rs:254
        u32le(&mut o, cs.blinding_factors());                   // verify.rs:281
        let pad = |o: &mut Vec<u8>| o.resize((o.len() + 3) & !3, 0);
        o.extend(cs.advice_column_phase.iter().map(|p| *p as u8)); // verify.rs:364 (phase as a small integer)
        pad(&mut o);
        o.extend(cs.challenge_phase.iter().map(|p| *p as u8));  // verify.rs:372
        pad(&mut o);
        for qs in [&cs.advice_queries.iter().map(|q| (q.0.index, q.1 .0)).collect::<Vec<_>>(),   // verify.rs:535-541
                   &cs.instance_queries.iter().map(|q| (q.0.index, q.1 .0)).collect::<Vec<_>>(), // verify.rs:524-530
                   &cs.fixed_queries.iter().map(|q| (q.0.index, q.1 .0)).collect::<Vec<_>>()] {  // verify.rs:544-550
            u32le(&mut o, qs.len());
            for (c, r) in qs.iter() {
                u32le(&mut o, *c);
                o.extend_from_slice(&(*r as i32).to_le_bytes());
            }
        }
        u32le(&mut o, cs.permutation.columns.len());            // verify.rs:249-268
        for col in cs.permutation.columns.iter() {
            u32le(&mut o, match col.column_type() { Any::Advice(_) => 0, Any::Fixed => 1, Any::Instance => 2 });
            u32le(&mut o, col.index());
        }
        u32le(&mut o, vk.fixed_commitments().len());            // verify.rs:476-481
        for p in vk.fixed_commitments() { affine_bytes(&mut o, p); }
        u32le(&mut o, vk.permutation().commitments.len());      // verify.rs:551-557
        for p in vk.permutation().commitments.iter() { affine_bytes(&mut o, p); }
        put_fe(&mut o, &vk_scalar);
        u32le(&mut o, cs.gates.len());                          // verify.rs:499-510
        for g in cs.gates.iter() { put_exprs(&mut o, &g.polys); }
        u32le(&mut o, cs.lookups.len());                        // verify.rs:313-322
        for l in cs.lookups.iter() {
            put_exprs(&mut o, &l.input_expressions);
            put_exprs(&mut o, &l.table_expressions);
        }
        o
    }

    /// calc_verify_circuit_final_pair (verify_circuit.rs:114-201) in one call: transcripts, expressions, multi_exps and the
    /// pairing on the GPU backend.  `circuits`: (vk blob, name, g_lagrange handle, per proof (instance columns, transcript)).
    pub fn final_pair(
        gpu: *mut h2agg_ctx,
        circuits: &[(Vec<u8>, String, u64, Vec<(Vec<Vec<u8>>, Vec<u8>)>)],
        s_g2: &[u8; 128], g2: &[u8; 128],
    ) -> ([u8; 64], [u8; 64], bool) {
        let mut vks = Vec::new();
        let mut keep: Vec<Box<dyn std::any::Any>> = Vec::new();
        let mut arr = Vec::new();
        for (blob, name, table, proofs) in circuits {
            let mut vk = std::ptr::null_mut();
            assert_eq!(unsafe { h2agg_vk_create(gpu, blob.as_ptr(), blob.len(), &mut vk) }, 0);
            vks.push(vk);
            let cname = std::ffi::CString::new(name.as_str()).unwrap();
            let tr: Vec<*const u8> = proofs.iter().map(|p| p.1.as_ptr()).collect();
            let tl: Vec<usize> = proofs.iter().map(|p| p.1.len()).collect();
            let flat: Vec<Vec<u8>> = proofs.iter().map(|p| p.0.concat()).collect();
            let ip: Vec<*const u8> = flat.iter().map(|v| v.as_ptr()).collect();
            let il: Vec<u32> = proofs.iter().flat_map(|p| p.0.iter().map(|c| (c.len() / 32) as u32)).collect();
            arr.push(h2agg_circuit_proofs { vk, name: cname.as_ptr(), g_lagrange: *table, nproofs: proofs.len(),
                transcripts: tr.as_ptr(), transcript_lens: tl.as_ptr(), instances: ip.as_ptr(), instance_lens: il.as_ptr() });
            keep.push(Box::new((cname, tr, tl, flat, ip, il)));
        }
        let (mut l, mut r, mut lam, mut ok) = ([0u8; 64], [0u8; 64], [0u8; 32], 0 as c_int);
        let rc = unsafe { h2agg_verify_aggregation(gpu, arr.as_ptr(), arr.len(), s_g2.as_ptr(), g2.as_ptr(), l.as_mut_ptr(), r.as_mut_ptr(), lam.as_mut_ptr(), &mut ok) };
        for vk in vks { unsafe { h2agg_vk_destroy(vk) } }
        assert_eq!(rc, 0, "h2agg_verify_aggregation failed"); // the Mock context panics where this returns an error
        (l, r, ok != 0)
    }
}
