// BN254 G1 group law for gfx950, device side.  y^2 = x^3 + 3 over Fq, generator (1, 2).
//
// Stands behind the halo2curves `G1` / `G1Affine` operators the reference's MockEccChip calls:
//   add / sub            halo2-snark-aggregator-api/src/mock/arith/ecc.rs:30-46   (`*a + *b`, `*a - *b`)
//   scalar_mul(_constant) halo2-snark-aggregator-api/src/mock/arith/ecc.rs:88-104 (`rhs * lhs`)
//   to_value             halo2-snark-aggregator-api/src/mock/arith/ecc.rs:64-66   (`to_affine`)
//
// Working representation is extended-Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity
// ZZ = 0): a mixed add costs 8M+2S and a full add 12M+2S, cheaper than Jacobian's 7M+4S / 11M+5S, and
// the accumulator of the Pippenger bucket loop lives in it.  The C ABI speaks the reference's own
// types: affine x||y (C = G1Affine) and Jacobian x||y||z (C::CurveExt); conversions are below.
// Every exceptional case of the group law (P = Q, P = -Q, identity operands) is handled explicitly —
// results must be bit-exact on adversarial inputs (duplicate bases, a base and its negation in one
// bucket), not only on random ones.
#pragma once
#include "fp.cuh"

namespace h2agg {

struct G1Affine {  // identity encoded as (0, 0), as halo2curves' G1Affine::identity()
    Fq x, y;
    FP_INLINE bool is_identity() const { return x.is_zero() && y.is_zero(); }
};
struct G1Jac {  // identity: z = 0
    Fq x, y, z;
};
struct G1XYZZ {  // identity: zz = 0
    Fq x, y, zz, zzz;
    FP_INLINE bool is_identity() const { return zz.is_zero(); }
    static FP_INLINE G1XYZZ identity() {
        G1XYZZ r;
        r.x = Fq::zero();
        r.y = Fq::one();
        r.zz = Fq::zero();
        r.zzz = Fq::zero();
        return r;
    }
    static FP_INLINE G1XYZZ from_affine(const G1Affine& p) {
        G1XYZZ r;
        if (p.is_identity()) return identity();
        r.x = p.x;
        r.y = p.y;
        r.zz = Fq::one();
        r.zzz = Fq::one();
        return r;
    }
};

#define FQ_MUL(a, b) fp_mul<FqParams>(a, b)
#define FQ_SQR(a) fp_sqr<FqParams>(a)
#define FQ_ADD(a, b) fp_add<FqParams>(a, b)
#define FQ_SUB(a, b) fp_sub<FqParams>(a, b)
#define FQ_DBL(a) fp_dbl<FqParams>(a)

// 2 * (affine point), mdbl-2008-s-1.  p must not be the identity; y = 0 cannot occur on a prime-order curve.
FP_INLINE G1XYZZ xyzz_double_affine(const G1Affine& p) {
    G1XYZZ r;
    Fq u = FQ_DBL(p.y);
    Fq v = FQ_SQR(u);
    Fq w = FQ_MUL(u, v);
    Fq s = FQ_MUL(p.x, v);
    Fq xx = FQ_SQR(p.x);
    Fq m = FQ_ADD(FQ_DBL(xx), xx);
    r.x = FQ_SUB(FQ_SQR(m), FQ_DBL(s));
    r.y = FQ_SUB(FQ_MUL(m, FQ_SUB(s, r.x)), FQ_MUL(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * p, dbl-2008-s-1 (a = 0)
FP_INLINE G1XYZZ xyzz_double(const G1XYZZ& p) {
    if (p.is_identity()) return p;
    G1XYZZ r;
    Fq u = FQ_DBL(p.y);
    Fq v = FQ_SQR(u);
    Fq w = FQ_MUL(u, v);
    Fq s = FQ_MUL(p.x, v);
    Fq xx = FQ_SQR(p.x);
    Fq m = FQ_ADD(FQ_DBL(xx), xx);
    r.x = FQ_SUB(FQ_SQR(m), FQ_DBL(s));
    r.y = FQ_SUB(FQ_MUL(m, FQ_SUB(s, r.x)), FQ_MUL(w, p.y));
    r.zz = FQ_MUL(v, p.zz);
    r.zzz = FQ_MUL(w, p.zzz);
    return r;
}

// acc += q (q affine), madd-2008-s, complete.
FP_INLINE void xyzz_add_affine(G1XYZZ& acc, const G1Affine& q) {
    if (q.is_identity()) return;
    if (acc.is_identity()) {
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = Fq::one();
        acc.zzz = Fq::one();
        return;
    }
    Fq u2 = FQ_MUL(q.x, acc.zz);
    Fq s2 = FQ_MUL(q.y, acc.zzz);
    Fq p = FQ_SUB(u2, acc.x);
    Fq r = FQ_SUB(s2, acc.y);
    if (p.is_zero()) {
        if (r.is_zero()) {
            acc = xyzz_double_affine(q);
        } else {
            acc = G1XYZZ::identity();
        }
        return;
    }
    Fq pp = FQ_SQR(p);
    Fq ppp = FQ_MUL(p, pp);
    Fq qq = FQ_MUL(acc.x, pp);
    Fq x3 = FQ_SUB(FQ_SUB(FQ_SQR(r), ppp), FQ_DBL(qq));
    Fq y3 = FQ_SUB(FQ_MUL(r, FQ_SUB(qq, x3)), FQ_MUL(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = FQ_MUL(acc.zz, pp);
    acc.zzz = FQ_MUL(acc.zzz, ppp);
}

// a + b, add-2008-s, complete.
FP_INLINE G1XYZZ xyzz_add(const G1XYZZ& a, const G1XYZZ& b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fq u1 = FQ_MUL(a.x, b.zz);
    Fq u2 = FQ_MUL(b.x, a.zz);
    Fq s1 = FQ_MUL(a.y, b.zzz);
    Fq s2 = FQ_MUL(b.y, a.zzz);
    Fq p = FQ_SUB(u2, u1);
    Fq r = FQ_SUB(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) return xyzz_double(a);
        return G1XYZZ::identity();
    }
    G1XYZZ o;
    Fq pp = FQ_SQR(p);
    Fq ppp = FQ_MUL(p, pp);
    Fq q = FQ_MUL(u1, pp);
    o.x = FQ_SUB(FQ_SUB(FQ_SQR(r), ppp), FQ_DBL(q));
    o.y = FQ_SUB(FQ_MUL(r, FQ_SUB(q, o.x)), FQ_MUL(s1, ppp));
    o.zz = FQ_MUL(FQ_MUL(a.zz, b.zz), pp);
    o.zzz = FQ_MUL(FQ_MUL(a.zzz, b.zzz), ppp);
    return o;
}

FP_INLINE G1XYZZ xyzz_neg(const G1XYZZ& a) {
    G1XYZZ r = a;
    r.y = fp_neg<FqParams>(a.y);
    return r;
}

// Jacobian (X, Y, Z) -> XYZZ: ZZ = Z^2, ZZZ = Z^3
FP_INLINE G1XYZZ xyzz_from_jac(const G1Jac& p) {
    G1XYZZ r;
    if (p.z.is_zero()) return G1XYZZ::identity();
    r.x = p.x;
    r.y = p.y;
    r.zz = FQ_SQR(p.z);
    r.zzz = FQ_MUL(r.zz, p.z);
    return r;
}
// XYZZ -> Jacobian with Z' = ZZZ: X' = X*ZZ^2, Y' = Y*ZZZ^2 (uses ZZ^3 = ZZZ^2).  Identity -> (0, 1, 0).
FP_INLINE G1Jac jac_from_xyzz(const G1XYZZ& p) {
    G1Jac r;
    if (p.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::one();
        r.z = Fq::zero();
        return r;
    }
    r.x = FQ_MUL(p.x, FQ_SQR(p.zz));
    r.y = FQ_MUL(p.y, FQ_SQR(p.zzz));
    r.z = p.zzz;
    return r;
}

// load / store helpers.  Device-resident points are Montgomery-form limbs; `canonical` variants convert
// from / to the C ABI's canonical little-endian integers.
FP_INLINE G1Affine affine_load(const void* p) {
    G1Affine r;
    r.x = fp_load<FqParams>(p);
    r.y = fp_load<FqParams>(reinterpret_cast<const uint8_t*>(p) + 32);
    return r;
}
FP_INLINE void affine_store(void* p, const G1Affine& a) {
    fp_store<FqParams>(p, a.x);
    fp_store<FqParams>(reinterpret_cast<uint8_t*>(p) + 32, a.y);
}
FP_INLINE G1XYZZ xyzz_load(const void* p) {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
    G1XYZZ r;
    r.x = fp_load<FqParams>(b);
    r.y = fp_load<FqParams>(b + 32);
    r.zz = fp_load<FqParams>(b + 64);
    r.zzz = fp_load<FqParams>(b + 96);
    return r;
}
FP_INLINE void xyzz_store(void* p, const G1XYZZ& a) {
    uint8_t* b = reinterpret_cast<uint8_t*>(p);
    fp_store<FqParams>(b, a.x);
    fp_store<FqParams>(b + 32, a.y);
    fp_store<FqParams>(b + 64, a.zz);
    fp_store<FqParams>(b + 96, a.zzz);
}
FP_INLINE G1Jac jac_load_canonical(const void* p) {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
    G1Jac r;
    r.x = fp_to_mont<FqParams>(fp_load<FqParams>(b));
    r.y = fp_to_mont<FqParams>(fp_load<FqParams>(b + 32));
    r.z = fp_to_mont<FqParams>(fp_load<FqParams>(b + 64));
    return r;
}
FP_INLINE void jac_store_canonical(void* p, const G1Jac& a) {
    uint8_t* b = reinterpret_cast<uint8_t*>(p);
    fp_store<FqParams>(b, fp_from_mont<FqParams>(a.x));
    fp_store<FqParams>(b + 32, fp_from_mont<FqParams>(a.y));
    fp_store<FqParams>(b + 64, fp_from_mont<FqParams>(a.z));
}

}  // namespace h2agg
