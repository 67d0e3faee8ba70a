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
#include "fp.hpp"

namespace h2agg {

#define FQ_MUL(a, b) fp_mul<FqParams>(a, b)
#define FQ_SQR(a) fp_sqr<FqParams>(a)
FP_INLINE void fq_fence(Fq& a) {   // no instructions: makes the limbs opaque 32-bit values again
#pragma unroll
    for (int i = 0; i < NL; ++i) asm("" : "+v"(a.l[i]));
}
#define FQ_ADD(a, b) fp_add<FqParams>(a, b)
#define FQ_SUB(K, a, b) fp_sub<K, FqParams>(a, b) /* a - b + K*p, needs value(b) <= K*p */
#define FQ_DBL(a) fp_dbl<FqParams>(a)

// Lazy-reduction invariant of every point held in registers or memory (bounds in multiples of p):
//     XYZZ:   X < 8,  Y < 4,  ZZ < 2,  ZZZ < 2          affine (Montgomery):  x, y <= 2
// Each formula is annotated [bound] so that every fp_mul / fp_sqr has (bound a)*(bound b) <= 169 (its
// result is then < 2) and every subtraction offset K >= the bound of its subtrahend.

struct G1Affine {  // identity encoded as the integers (0, 0), as halo2curves' G1Affine::identity()
    Fq x, y;
    FP_INLINE bool is_identity() const { return x.is_zero_int() && y.is_zero_int(); }
};
struct G1Jac {  // identity: z = 0
    Fq x, y, z;
};
struct G1XYZZ {  // identity: zz is the integer 0 (a computed ZZ of a finite point is never 0 mod p)
    Fq x, y, zz, zzz;
    FP_INLINE bool is_identity() const { return zz.is_zero_int(); }
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

// -q for an affine point (identity stays the identity)
FP_INLINE G1Affine affine_neg(const G1Affine& q) {
    G1Affine r = q;
    if (!q.is_identity()) r.y = fp_neg<2, FqParams>(q.y);  // [<= 2]
    return r;
}

// shared tail of the doubling formulas: given U = 2Y [u<=8], X [<8], Y [<4] returns X3, Y3 and V, W
FP_INLINE void xyzz_double_core(const Fq& x, const Fq& y, Fq& x3, Fq& y3, Fq& v, Fq& w) {
    Fq u = FQ_DBL(y);                                   // [8]
    v = FQ_SQR(u);                                      // 64  -> [2]
    w = FQ_MUL(u, v);                                   // 16  -> [2]
    Fq s = FQ_MUL(x, v);                                // 16  -> [2]
    Fq xx = FQ_SQR(x);                                  // 64  -> [2]
    Fq m = fp_triple<FqParams>(xx);                     // [6]
    x3 = fp_sub2<4, FqParams>(FQ_SQR(m), s);            // 36 -> [2]; 2S [4]  -> [6]
    // M*(S - X3) - W*Y as ONE reduction: M*(S - X3 + 6p) + W*(4p - Y):  (6*8 + 2*4)/169 + 1 -> [2]
    y3 = fp_mul2<FqParams>(m, FQ_SUB(6, s, x3), w, fp_neg<4, FqParams>(y));
}

// 2 * (affine point), mdbl-2008-s-1.  p must not be the identity; y = 0 cannot occur on a prime-order curve.
FP_INLINE G1XYZZ xyzz_double_affine(const G1Affine& p) {
    G1XYZZ r;
    xyzz_double_core(p.x, p.y, r.x, r.y, r.zz, r.zzz);
    return r;
}

// 2 * p, dbl-2008-s-1 (a = 0)
FP_INLINE G1XYZZ xyzz_double(const G1XYZZ& p) {
    if (p.is_identity()) return p;
    G1XYZZ r;
    Fq v, w;
    xyzz_double_core(p.x, p.y, r.x, r.y, v, w);
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
#ifndef H2AGG_SINGLE_MUL   // (A/B switch: -DH2AGG_SINGLE_MUL restores one product at a time)
    Fq u2, s2;
    fp_mul_dual<FqParams>(q.x, acc.zz, q.y, acc.zzz, u2, s2);   // 2*2 -> [2], [2]
#else
    Fq u2 = FQ_MUL(q.x, acc.zz);                        // 2*2 -> [2]
    Fq s2 = FQ_MUL(q.y, acc.zzz);                       // [2]
#endif
    Fq p = FQ_SUB(8, u2, acc.x);                        // [10]
    Fq r = FQ_SUB(4, s2, acc.y);                        // [6]
    if (fp_maybe_zero_mod<10, FqParams>(p)) {           // exact "no"; the rare "maybe" is decided exactly
        if (fp_is_zero_mod<10, FqParams>(p)) {
            if (fp_is_zero_mod<6, FqParams>(r)) {
                acc = xyzz_double_affine(q);
            } else {
                acc = G1XYZZ::identity();
            }
            return;
        }
    }
    // Optimiser fence on the limbs that live across the exceptional-case branch above: without it LLVM carries them as
    // zero-extended 64-bit values through the branch's merge point and then multiplies 64 x 32 bits (an extra mad with a
    // zero high word, a v_mul_lo / v_add3 and register shuffles per product): -7 % instructions in the mixed addition.
    fq_fence(p);
    fq_fence(r);
#ifndef H2AGG_SINGLE_MUL
    // independent products side by side, one multiply-add chain each (fp_mont_chain2): (PP, RR), (PPP, Q), (Y3, ZZ3, ZZZ3)
    Fq pp, rr, ppp, qq;
    fp_sqr_dual<FqParams>(p, r, pp, rr);                // 100 -> [2], 36 -> [2]
    fp_mul_dual<FqParams>(p, pp, acc.x, pp, ppp, qq);   // 20 -> [2], 16 -> [2]
    Fq x3 = fp_sub_sub2<6, FqParams>(rr, ppp, qq);      // PPP + 2Q [6] -> [8]
    // R*(Q - X3) - Y1*PPP as ONE reduction: R*(Q - X3 + 8p) + (4p - Y1)*PPP: (6*10 + 4*2)/169 + 1 -> [2]
    Fq y3, zz3, zzz3;
    fp_mul2_mul_mul<FqParams>(r, FQ_SUB(8, qq, x3), fp_neg<4, FqParams>(acc.y), ppp, acc.zz, pp, acc.zzz, ppp, y3, zz3, zzz3);
    acc.x = x3;
    acc.y = y3;
    acc.zz = zz3;
    acc.zzz = zzz3;
#else
    Fq pp = FQ_SQR(p);                                  // 100 -> [2]
    Fq ppp = FQ_MUL(p, pp);                             // 20  -> [2]
    Fq qq = FQ_MUL(acc.x, pp);                          // 16  -> [2]
    Fq x3 = fp_sub_sub2<6, FqParams>(FQ_SQR(r), ppp, qq);           // 36 -> [2]; PPP + 2Q [6] -> [8]
    // R*(Q - X3) - Y1*PPP as ONE reduction: R*(Q - X3 + 8p) + (4p - Y1)*PPP: (6*10 + 4*2)/169 + 1 -> [2]
    Fq y3 = fp_mul2<FqParams>(r, FQ_SUB(8, qq, x3), fp_neg<4, FqParams>(acc.y), ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = FQ_MUL(acc.zz, pp);                        // [2]
    acc.zzz = FQ_MUL(acc.zzz, ppp);                     // [2]
#endif
}

// a + q for two AFFINE points (mmadd-2008-s, 4M + 2S against the 8M + 2S of a mixed addition): the second point that
// lands in a bucket.  Neither operand is the identity.
FP_INLINE G1XYZZ xyzz_add_affine_affine(const G1Affine& a, const G1Affine& q) {
    Fq p = FQ_SUB(2, q.x, a.x);                         // [4]
    Fq r = FQ_SUB(2, q.y, a.y);                         // [4]
    if (fp_maybe_zero_mod<4, FqParams>(p)) {
        if (fp_is_zero_mod<4, FqParams>(p)) {
            if (fp_is_zero_mod<4, FqParams>(r)) return xyzz_double_affine(q);
            return G1XYZZ::identity();
        }
    }
    fq_fence(p);
    fq_fence(r);
    G1XYZZ o;
    Fq pp, rr, ppp, qq;
    fp_sqr_dual<FqParams>(p, r, pp, rr);                // 16 -> [2], [2]
    fp_mul_dual<FqParams>(p, pp, a.x, pp, ppp, qq);     // [2], [2]
    o.x = fp_sub_sub2<6, FqParams>(rr, ppp, qq);        // [8]
    // R*(Q - X3 + 8p) + (2p - Y1)*PPP: (4*10 + 2*2)/169 + 1 -> [2]
    o.y = fp_mul2<FqParams>(r, FQ_SUB(8, qq, o.x), fp_neg<2, FqParams>(a.y), ppp);
    o.zz = pp;
    o.zzz = ppp;
    return o;
}

// a + b, add-2008-s, complete.
FP_INLINE G1XYZZ xyzz_add(const G1XYZZ& a, const G1XYZZ& b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fq u1 = FQ_MUL(a.x, b.zz);                          // 16 -> [2]
    Fq u2 = FQ_MUL(b.x, a.zz);                          // [2]
    Fq s1 = FQ_MUL(a.y, b.zzz);                         // 8 -> [2]
    Fq s2 = FQ_MUL(b.y, a.zzz);                         // [2]
    Fq p = FQ_SUB(2, u2, u1);                           // [4]
    Fq r = FQ_SUB(2, s2, s1);                           // [4]
    if (fp_maybe_zero_mod<4, FqParams>(p)) {
        if (fp_is_zero_mod<4, FqParams>(p)) {
            if (fp_is_zero_mod<4, FqParams>(r)) return xyzz_double(a);
            return G1XYZZ::identity();
        }
    }
    G1XYZZ o;
    fq_fence(p);   // see xyzz_add_affine
    fq_fence(r);
    Fq pp = FQ_SQR(p);                                  // 16 -> [2]
    Fq ppp = FQ_MUL(p, pp);                             // [2]
    Fq q = FQ_MUL(u1, pp);                              // [2]
    o.x = fp_sub_sub2<6, FqParams>(FQ_SQR(r), ppp, q);                   // [8]
    // R*(Q - X3 + 8p) + (2p - S1)*PPP: (4*10 + 2*2)/169 + 1 -> [2]
    o.y = fp_mul2<FqParams>(r, FQ_SUB(8, q, o.x), fp_neg<2, FqParams>(s1), ppp);
    o.zz = FQ_MUL(FQ_MUL(a.zz, b.zz), pp);              // [2]
    o.zzz = FQ_MUL(FQ_MUL(a.zzz, b.zzz), ppp);          // [2]
    return o;
}

// a + b as xyzz_add, with the multiplications as side-by-side chains (fp_mont_chain2): fewer instructions (no merges of
// partial sums), but one dependent chain per product — for kernels that have other waves to fill the wait states
// (the row / column sums of the bucket reduction), not for the lone-wave tails.
FP_INLINE G1XYZZ xyzz_add_chains(const G1XYZZ& a, const G1XYZZ& b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fq u1, u2, s1, s2;
    fp_mul_dual<FqParams>(a.x, b.zz, b.x, a.zz, u1, u2);        // 16 -> [2], [2]
    fp_mul_dual<FqParams>(a.y, b.zzz, b.y, a.zzz, s1, s2);      // 8 -> [2], [2]
    Fq p = FQ_SUB(2, u2, u1);                                   // [4]
    Fq r = FQ_SUB(2, s2, s1);                                   // [4]
    if (fp_maybe_zero_mod<4, FqParams>(p)) {
        if (fp_is_zero_mod<4, FqParams>(p)) {
            if (fp_is_zero_mod<4, FqParams>(r)) return xyzz_double(a);
            return G1XYZZ::identity();
        }
    }
    G1XYZZ o;
    fq_fence(p);
    fq_fence(r);
    Fq pp, rr, ppp, q, zz12, zzz12;
    fp_sqr_dual<FqParams>(p, r, pp, rr);                        // 16 -> [2], [2]
    fp_mul_dual<FqParams>(a.zz, b.zz, a.zzz, b.zzz, zz12, zzz12);
    fp_mul_dual<FqParams>(p, pp, u1, pp, ppp, q);               // [2], [2]
    o.x = fp_sub_sub2<6, FqParams>(rr, ppp, q);                 // [8]
    // R*(Q - X3 + 8p) + (2p - S1)*PPP: (4*10 + 2*2)/169 + 1 -> [2]
    fp_mul2_mul_mul<FqParams>(r, FQ_SUB(8, q, o.x), fp_neg<2, FqParams>(s1), ppp, zz12, pp, zzz12, ppp, o.y, o.zz, o.zzz);
    return o;
}

FP_INLINE G1XYZZ xyzz_neg(const G1XYZZ& a) {
    G1XYZZ r = a;
    if (!a.is_identity()) r.y = fp_neg<4, FqParams>(a.y);  // [4]
    return r;
}

// Jacobian (X, Y, Z) [each <= 2] -> XYZZ: ZZ = Z^2, ZZZ = Z^3.  Identity <=> Z = 0 mod p.
FP_INLINE G1XYZZ xyzz_from_jac(const G1Jac& p) {
    G1XYZZ r;
    if (fp_is_zero_mod<2, FqParams>(p.z)) return G1XYZZ::identity();
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
    r.x = FQ_MUL(p.x, FQ_SQR(p.zz));                    // 8*2 -> [2]
    r.y = FQ_MUL(p.y, FQ_SQR(p.zzz));                   // [2]
    r.z = p.zzz;
    return r;
}

// ---- memory formats ---------------------------------------------------------------------------------
// base tables:   64 B / point, x || y, each a packed 256-bit Montgomery value (< 2p)
// XYZZ records: 144 B / point (internal: buckets, partial sums), 4 x 9 limbs as 9 x uint4
constexpr int XYZZ_BYTES = 144;

FP_INLINE G1Affine affine_load(const void* p) {
    G1Affine r;
    r.x = fp_load<FqParams>(p);
    r.y = fp_load<FqParams>(reinterpret_cast<const uint8_t*>(p) + 32);
    return r;
}
FP_INLINE void affine_store(void* p, const G1Affine& a) {  // coordinates must be < 2^256 (Montgomery outputs are)
    fp_store<FqParams>(p, a.x);
    fp_store<FqParams>(reinterpret_cast<uint8_t*>(p) + 32, a.y);
}
FP_INLINE G1XYZZ xyzz_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint4 v = q[i];
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
    G1XYZZ r;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        r.x.l[i] = w[i];
        r.y.l[i] = w[9 + i];
        r.zz.l[i] = w[18 + i];
        r.zzz.l[i] = w[27 + i];
    }
    return r;
}
FP_INLINE void xyzz_store(void* p, const G1XYZZ& a) {
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        w[i] = a.x.l[i];
        w[9 + i] = a.y.l[i];
        w[18 + i] = a.zz.l[i];
        w[27 + i] = a.zzz.l[i];
    }
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 9; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// C-ABI Jacobian: three canonical little-endian integers
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
