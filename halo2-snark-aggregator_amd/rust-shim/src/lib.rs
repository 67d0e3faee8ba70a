//! FFI + marshalling for libh2agg.so, the MI355X backend of halo2-snark-aggregator's pure-calculation context.
//!
//! SOURCE ONLY — never compiled in the build image (no Rust toolchain there).  Layout facts about halo2curves 0.2.1 used
//! below (`to_repr()` = 32-byte little-endian canonical integer, `jacobian_coordinates()` = (x, y, z) of a `CurveExt`,
//! identity = z == 0) must be re-verified where Rust exists.
//!
//! How it is used: `integration/api-h2agg.patch` gives halo2-snark-aggregator-api a feature `h2agg`; under it the body of
//! `MockEccChip::multi_exp` (mock/arith/ecc.rs:106-129) is `multi_exp_with_point_list` below.  The chip's TYPE, its
//! associated types (`AssignedPoint = C::CurveExt`) and every other method stay the reference's, so -circuit
//! (verify_circuit.rs:41-45,115-118) and -sdk compile untouched.
use group::Group;
use halo2_proofs::arithmetic::{CurveAffine, CurveExt, FieldExt};
use std::cell::RefCell;
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct h2agg_ctx {
    _private: [u8; 0],
}

extern "C" {
    // include/h2agg.h
    pub fn h2agg_create(device_ordinal: c_int, out: *mut *mut h2agg_ctx) -> c_int;
    pub fn h2agg_destroy(ctx: *mut h2agg_ctx);
    pub fn h2agg_last_error(ctx: *const h2agg_ctx) -> *const c_char;
    pub fn h2agg_g1_msm(ctx: *mut h2agg_ctx, bases_aff: *const u8, scalars: *const u8, n: usize, out_jac: *mut u8) -> c_int;
    pub fn h2agg_g1_msm_jac(ctx: *mut h2agg_ctx, points_jac: *const u8, scalars: *const u8, n: usize, out_jac: *mut u8) -> c_int;
    pub fn h2agg_g1_batch_scalar_mul(ctx: *mut h2agg_ctx, bases_aff: *const u8, scalars: *const u8, n: usize, out_jac: *mut u8) -> c_int;
    pub fn h2agg_host_alloc(ctx: *mut h2agg_ctx, bytes: usize, out: *mut *mut u8) -> c_int;
    pub fn h2agg_host_free(ctx: *mut h2agg_ctx, p: *mut u8) -> c_int;
}

/// One context per thread (the chips are used single-threaded, verify_circuit.rs:114-201; the SDK's rayon workers each get
/// their own) with its page-locked marshalling buffers: points and scalars are written STRAIGHT into memory from
/// h2agg_host_alloc, so the 128 B per pair cross PCIe at link rate, sliced under the device's compute — a pageable `Vec`
/// costs 4.6 ms per 2^20 pairs instead of 2.9.
pub struct Gpu {
    pub ctx: *mut h2agg_ctx,
    points: *mut u8,
    scalars: *mut u8,
    cap_pairs: usize,
}
impl Gpu {
    fn new() -> Self {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { h2agg_create(0, &mut ctx) };
        assert_eq!(rc, 0, "h2agg_create failed: a HIP device is required (there is no CPU mode)");
        Gpu { ctx, points: std::ptr::null_mut(), scalars: std::ptr::null_mut(), cap_pairs: 0 }
    }
    /// page-locked room for n pairs (grow-only, doubling)
    fn reserve(&mut self, n: usize) {
        if n <= self.cap_pairs {
            return;
        }
        let cap = n.next_power_of_two().max(1024);
        unsafe {
            if !self.points.is_null() {
                h2agg_host_free(self.ctx, self.points);
                h2agg_host_free(self.ctx, self.scalars);
            }
            assert_eq!(h2agg_host_alloc(self.ctx, 96 * cap, &mut self.points), 0, "h2agg_host_alloc");
            assert_eq!(h2agg_host_alloc(self.ctx, 32 * cap, &mut self.scalars), 0, "h2agg_host_alloc");
        }
        self.cap_pairs = cap;
    }
}
impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe {
            if !self.points.is_null() {
                h2agg_host_free(self.ctx, self.points);
                h2agg_host_free(self.ctx, self.scalars);
            }
            h2agg_destroy(self.ctx)
        }
    }
}
thread_local! {
    static GPU: RefCell<Option<Gpu>> = RefCell::new(None);
}
/// run `f` with this thread's context (created on first use)
pub fn with_gpu<R>(f: impl FnOnce(&mut Gpu) -> R) -> R {
    GPU.with(|g| f(g.borrow_mut().get_or_insert_with(Gpu::new)))
}

#[inline]
fn write_fe<F: FieldExt>(dst: *mut u8, f: &F) {
    // 32-byte little-endian canonical integer
    unsafe { std::ptr::copy_nonoverlapping(f.to_repr().as_ref().as_ptr(), dst, 32) }
}
pub fn put_fe<F: FieldExt>(dst: &mut Vec<u8>, f: &F) {
    dst.extend_from_slice(f.to_repr().as_ref());
}
pub fn affine_bytes<C: CurveAffine>(dst: &mut Vec<u8>, p: &C) {
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
/// Jacobian (x, y, z) from the library -> C::CurveExt, via the affine point (one Fq inversion on the host, the same
/// `to_affine` the reference performs at verify_circuit.rs:180,200).
pub fn curve_from_jac<C: CurveAffine>(b: &[u8; 96]) -> C::CurveExt {
    let (x, y, z): (C::Base, C::Base, C::Base) = (fe_from(&b[0..32]), fe_from(&b[32..64]), fe_from(&b[64..96]));
    if bool::from(z.is_zero()) {
        return C::CurveExt::identity();
    }
    let zi = z.invert().unwrap();
    let zi2 = zi.square();
    C::from_xy(x * zi2, y * zi2 * zi).unwrap().into()
}

/// write pairs [lo, hi) into the page-locked buffers (x || y || z and the scalar, 32-byte LE canonical each)
fn marshal_range<C: CurveAffine>(points: &[C::CurveExt], scalars: &[C::ScalarExt], lo: usize, hi: usize, pb: SendPtr, sb: SendPtr) {
    for i in lo..hi {
        let (x, y, z) = points[i].jacobian_coordinates();
        unsafe {
            write_fe(pb.0.add(96 * i), &x);
            write_fe(pb.0.add(96 * i + 32), &y);
            write_fe(pb.0.add(96 * i + 64), &z);
            write_fe(sb.0.add(32 * i), &scalars[i]);
        }
    }
}
#[derive(Clone, Copy)]
struct SendPtr(*mut u8);
unsafe impl Send for SendPtr {} // disjoint ranges of one page-locked buffer, written by scoped threads
unsafe impl Sync for SendPtr {}

fn run_msm<C: CurveAffine>(g: &mut Gpu, n: usize) -> C::CurveExt {
    let mut out = [0u8; 96];
    let rc = unsafe { h2agg_g1_msm_jac(g.ctx, g.points, g.scalars, n, out.as_mut_ptr()) };
    if rc != 0 {
        // H2AGG_ERR_EMPTY (3) reproduces the reference's `acc.unwrap()` panic on zero pairs
        let msg = unsafe { std::ffi::CStr::from_ptr(h2agg_last_error(g.ctx)) };
        panic!("h2agg_g1_msm_jac failed ({}): {:?}", rc, msg);
    }
    curve_from_jac::<C>(&out)
}

/// sum_i scalars[i] * points[i]  (MockEccChip::multi_exp, mock/arith/ecc.rs:106-129, without its `point_list` side effect).
/// The points go over as they are — projective x || y || z — and are normalised on the device (h2agg_g1_msm_jac):
/// `batch_normalize` over 2^20 points is a third of a second on one host core, the whole MSM is ~3.6 ms from host buffers.
pub fn multi_exp<C: CurveAffine>(points: &[C::CurveExt], scalars: &[C::ScalarExt]) -> C::CurveExt {
    let n = points.len().min(scalars.len()); // `zip` semantics of the reference's loop
    with_gpu(|g| {
        g.reserve(n.max(1));
        marshal_range::<C>(points, scalars, 0, n, SendPtr(g.points), SendPtr(g.scalars));
        run_msm::<C>(g, n)
    })
}

/// `multi_exp` plus the reference's observable side effect: `ctx.point_list = points.map(|x| format!("{:?}", x))`
/// (mock/arith/ecc.rs:112-116; `MockChipCtx::point_list` is a public field, `Display` prints its length).  Formatting 2^20
/// points is 0.3-0.6 s on one core — 100x the device's work — so scoped worker threads take one contiguous chunk each:
/// first they marshal it into the page-locked buffers (33 ms on one thread otherwise), then they format it, WHILE this
/// thread, once every chunk is marshalled, runs the GPU call.  The strings and their order are exactly the reference's.
/// Measured with the C++ stand-in tools/dropin_cost.cpp on the GPU box (16 cores' quota): 196 ms end to end per 2^20
/// pairs against 336 ms in the reference's order — the list, not the multi_exp, is what a drop-in caller waits for.
pub fn multi_exp_with_point_list<C: CurveAffine>(
    points: &[C::CurveExt],
    scalars: &[C::ScalarExt],
    point_list: &mut Vec<String>,
) -> C::CurveExt
where
    C::CurveExt: Sync,
    C::ScalarExt: Sync,
{
    let n = points.len().min(scalars.len());
    // Small multi_exps — the ~350- and ~1 400-pair ones of an evaluation, every MockEccChip::multi_exp under the h2agg feature
    // comes through here — stay on the calling thread: spawning and joining a thread per core costs tens of microseconds
    // each, more than marshalling and formatting a few thousand points.
    const INLINE_BELOW: usize = 1 << 14;
    if points.len() < INLINE_BELOW {
        let result = with_gpu(|g| {
            g.reserve(n.max(1));
            marshal_range::<C>(points, scalars, 0, n, SendPtr(g.points), SendPtr(g.scalars));
            run_msm::<C>(g, n)
        });
        point_list.clear();
        point_list.extend(points.iter().map(|x| format!("{:?}", x)));
        return result;
    }
    let workers = std::thread::available_parallelism().map(|v| v.get()).unwrap_or(1).saturating_sub(1).max(1);
    let chunk = ((points.len() + workers - 1) / workers).max(1);
    let nchunks = (points.len() + chunk - 1) / chunk;
    let mut parts: Vec<Vec<String>> = Vec::new();
    let result = with_gpu(|g| {
        g.reserve(n.max(1));
        let (pb, sb) = (SendPtr(g.points), SendPtr(g.scalars));
        std::thread::scope(|s| {
            // one channel message per marshalled chunk instead of a Barrier: a worker that panics before it reports (out of
            // memory inside marshal_range, say) drops its sender, `recv` then fails on this thread and the panic is
            // re-raised by the scope's join — with a Barrier this thread would have waited for it forever
            let (tx, rx) = std::sync::mpsc::channel::<()>();
            let handles: Vec<_> = (0..nchunks)
                .map(|w| {
                    let tx = tx.clone();
                    s.spawn(move || {
                        let (lo, hi) = (w * chunk, ((w + 1) * chunk).min(points.len()));
                        marshal_range::<C>(points, scalars, lo.min(n), hi.min(n), pb, sb);
                        let _ = tx.send(());
                        drop(tx);
                        points[lo..hi].iter().map(|x| format!("{:?}", x)).collect::<Vec<String>>()
                    })
                })
                .collect();
            drop(tx);
            let mut marshalled = 0;
            while marshalled < nchunks {
                match rx.recv() {
                    Ok(()) => marshalled += 1,
                    Err(_) => break, // every sender is gone and a chunk is missing: a worker panicked; the joins below re-raise it
                }
            }
            let r = if marshalled == nchunks { Some(run_msm::<C>(g, n)) } else { None }; // the GPU call, under the formatting
            parts = handles.into_iter().map(|h| h.join().unwrap_or_else(|e| std::panic::resume_unwind(e))).collect();
            r.expect("every chunk marshalled")
        })
    });
    point_list.clear();
    point_list.reserve(points.len());
    for p in parts {
        point_list.extend(p);
    }
    result
}

// =====================================================================================================================
// The entry points ABOVE multi_exp — what `calc_verify_circuit_final_pair`
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
    /// include/h2agg.h `h2agg_shard`: this rank's place in a sharded aggregation; `allgather` = the host's transport
    /// (None: the context's RCCL communicator, h2agg_comm_init_rank)
    #[repr(C)]
    pub struct h2agg_shard {
        pub rank: u32,
        pub world: u32,
        pub total_proofs: usize,
        pub global_index: *const u32,
        pub allgather: Option<unsafe extern "C" fn(user: *mut std::ffi::c_void, send: *const std::ffi::c_void, bytes: usize, recv: *mut std::ffi::c_void) -> c_int>,
        pub user: *mut std::ffi::c_void,
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
        // the same with the proofs sharded over ranks: both exchanges (every proof's squeeze -> the same lambda everywhere;
        // the partial pairs -> their sum) happen inside the call
        pub fn h2agg_verify_aggregation_sharded(
            ctx: *mut h2agg_ctx, circuits: *const h2agg_circuit_proofs, ncircuits: usize, shard: *const h2agg_shard,
            s_g2: *const u8, g2: *const u8, left_aff: *mut u8, right_aff: *mut u8, lambda_out: *mut u8, pairing_ok: *mut c_int,
            advice_out: *mut u8, advice_cap: usize,
        ) -> c_int;
        // one process per GPU: rank 0 makes the id, every rank joins
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
    /// reads it through.  `vk_scalar` = the value init_transcript absorbs (verify.rs:57-70).  FIELD ORDER (the decoder is
    /// csrc/verifier.inc `decode_vk`; the Python encoder halo2-snark-aggregator_amd/verifier.py::encode_vk writes the same
    /// and is round-tripped against it in tests/test_verifier_pipeline.py::test_h2vk_blob_field_order):
    ///   magic "H2VK", version 1, k, num_advice_columns, num_instance_columns, num_challenges, degree, blinding_factors,
    ///   advice_column_phase (bytes, padded to 4), challenge_phase (bytes, padded to 4),
    ///   advice / instance / fixed queries (count, then (column u32, rotation i32) each),
    ///   permutation columns (count, then (kind 0 advice | 1 fixed | 2 instance, index) each),
    ///   fixed commitments (count, 64 B each), permutation commitments (count, 64 B each), vk_scalar (32 B),
    ///   gates (count; per gate its polynomials as expressions), lookups (count; per lookup input then table expressions).
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
