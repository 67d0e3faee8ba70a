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
        let mut affine = vec![C::identity(); points.len()];
        C::CurveExt::batch_normalize(&points, &mut affine); // one shared inversion
        let (mut pb, mut sb) = (Vec::with_capacity(64 * n), Vec::with_capacity(32 * n));
        for i in 0..n {
            affine_bytes(&mut pb, &affine[i]);
            put_fe(&mut sb, &scalars[i]);
        }
        let mut out = [0u8; 96];
        let rc = unsafe { h2agg_g1_msm(self.gpu, pb.as_ptr(), sb.as_ptr(), n, out.as_mut_ptr()) };
        if rc != 0 {
            // H2AGG_ERR_EMPTY (3) reproduces the reference's `acc.unwrap()` panic on zero pairs
            let msg = unsafe { std::ffi::CStr::from_ptr(h2agg_last_error(self.gpu)) };
            panic!("h2agg_g1_msm failed ({}): {:?}", rc, msg);
        }
        Ok(curve_from_jac::<C>(&out))
    }
}
