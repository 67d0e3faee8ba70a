// BN254 prime-field arithmetic for gfx950 (CDNA4), device side.
//
// Stands behind the halo2curves `bn256::{Fq,Fr}` operators the reference calls from
//   halo2-snark-aggregator-api/src/mock/arith/field.rs:45,54,104,113,121,132,144   (Fr: add/sub/mul/invert)
//   halo2-snark-aggregator-api/src/mock/arith/ecc.rs:36,45,94,103                  (Fq inside every G1 op)
//
// Representation (round-1 measurement, profiles/r01_ubench_instruction_rates.txt: v_mad_u64_u32 issues
// at HALF rate on gfx950 — 4 cycles per wave64, same as v_fma_f64 / v_lshl_add_u64 — while a saturated
// 8x32-bit CIOS spends 2/3 of its cycles on carry bookkeeping): a field element is NINE 29-bit limbs in
// VGPRs ("unsaturated" radix 2^29, 261 bits for 254-bit moduli).  A column of the schoolbook product
// is a plain chain of v_mad_u64_u32 into one 64-bit accumulator — 18 products of < 2^58 never
// overflow it — so a Montgomery multiplication (R = 2^261) is 162 mads + one carry sweep, with no carry
// flags (VCC) anywhere and nine independent chains of ILP.
//
// Values are kept *lazily reduced*: limbs are tight (limbs 0..7 < 2^29), but the integer may be any
// small multiple range [0, B*m).  With R/m ~ 2^7.4 = 169 a product of inputs < A*m and < B*m comes out
// < (A*B/169 + 1)*m, so bounds do not grow through chains of multiplications; every function below
// states the bound it needs / yields.  Subtraction adds K*m first (template parameter K >= bound of the
// subtrahend) and carry-normalises with arithmetic shifts.  Exact zero / equality tests (needed for the
// exceptional cases of the group law — results must be bit-exact) use a one-limb filter that cannot
// miss a multiple of m, followed by a full reduction only when the filter fires.  No MFMA: this is
// integer arithmetic on the VALU, not a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FP_INLINE __device__ __forceinline__

namespace h2agg {

constexpr int NL = 9;                       // limbs
constexpr uint32_t M29 = (1u << 29) - 1u;   // limb mask

struct FqParams {
    // p = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    static constexpr uint32_t MOD[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                        0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x04866389u;  // -p^-1 mod 2^29
    static constexpr uint32_t R1[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                       0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
    static constexpr uint32_t R2[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u,
                                       0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};  // 2^522 mod p
};
struct FrParams {
    // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    static constexpr uint32_t MOD[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                        0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x0fffffffu;  // -r^-1 mod 2^29
    static constexpr uint32_t R1[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                       0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod r
    static constexpr uint32_t R2[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                       0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};  // 2^522 mod r
};

// i-th tight limb of K * modulus (compile-time; K <= 64 keeps the value < 2^261)
template <class P>
constexpr uint32_t km_limb(int K, int i) {
    uint64_t c = 0;
    uint32_t out = 0;
    for (int j = 0; j <= i; ++j) {
        uint64_t t = (uint64_t)P::MOD[j] * (uint64_t)K + c;
        out = (j < 8) ? (uint32_t)(t & M29) : (uint32_t)t;
        c = t >> 29;
    }
    return out;
}

template <class P>
struct Fp {
    uint32_t l[NL];

    static FP_INLINE Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.l[i] = 0;
        return r;
    }
    static FP_INLINE Fp one() {  // Montgomery 1 (< m)
        Fp r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.l[i] = P::R1[i];
        return r;
    }
    static FP_INLINE Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.l[i] = P::R2[i];
        return r;
    }
    // all limbs zero (integer zero) — NOT "zero mod m"; see fp_is_zero_mod
    FP_INLINE bool is_zero_int() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) o |= l[i];
        return o == 0;
    }
    FP_INLINE bool same_int(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) o |= (l[i] ^ b.l[i]);
        return o == 0;
    }
};

// carry-normalise signed limb values x[i] (|x[i]| < 2^31, total value in [0, 2^261)) to tight limbs
template <class P>
FP_INLINE Fp<P> fp_normalize(const int32_t (&x)[NL]) {
    Fp<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int32_t t = x[i] + c;
        r.l[i] = (uint32_t)t & M29;
        c = t >> 29;  // arithmetic
    }
    r.l[8] = (uint32_t)(x[8] + c);
    return r;
}

// a + b.  value bound: A + B.
template <class P>
FP_INLINE Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)(a.l[i] + b.l[i]);
    return fp_normalize<P>(x);
}
// 2a.  bound: 2A.
template <class P>
FP_INLINE Fp<P> fp_dbl(const Fp<P>& a) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)(a.l[i] << 1);
    return fp_normalize<P>(x);
}
// a - b + K*m.  REQUIRES value(b) <= K*m.  bound: A + K.
template <int K, class P>
FP_INLINE Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)(a.l[i] + km_limb<P>(K, i)) - (int32_t)b.l[i];
    return fp_normalize<P>(x);
}
// K*m - a.  REQUIRES value(a) <= K*m.  bound: K.
template <int K, class P>
FP_INLINE Fp<P> fp_neg(const Fp<P>& a) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)km_limb<P>(K, i) - (int32_t)a.l[i];
    return fp_normalize<P>(x);
}

// Montgomery reduction of an 18-column accumulator (columns < 2^63) -> tight limbs, value < (T/R + 1)*m
template <class P>
FP_INLINE Fp<P> fp_mont_reduce(uint64_t (&acc)[18]) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const uint32_t m = ((uint32_t)acc[k] * P::NINV) & M29;
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[k + j] += (uint64_t)m * P::MOD[j];
        acc[k + 1] += acc[k] >> 29;  // low 29 bits of acc[k] are now zero
    }
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = acc[9 + i] + c;
        r.l[i] = (uint32_t)t & M29;
        c = t >> 29;
    }
    r.l[8] = (uint32_t)(acc[17] + c);
    return r;
}

// Product-scanning (column-wise, "FIPS") Montgomery reduction engine: one running pair of 64-bit accumulators
// instead of the 18-column array of the operand-scanning form — same multiply-add count, ~30 fewer live VGPRs
// (the accumulate kernel drops under the 128-VGPR line = 4 waves per SIMD instead of 3).
// `col(k, t, u)` adds the product terms of column k (k = 0..16) into t / u; column k then also collects
// m_i*p_(k-i), m_k makes it divisible by 2^29 and the quotient carries into column k+1.  At most 27 products
// < 2^58 plus a carry < 2^36 per column: no overflow.
template <class P, class ColF>
FP_INLINE Fp<P> fp_mont_ps(ColF&& col) {
    uint32_t m[NL];
    Fp<P> r;
    uint64_t t = 0, u = 0;  // two interleaved partial sums halve the dependent chain
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        col(k, t, u);
#pragma unroll
        for (int i = 0; i < k; ++i) {
            if (i & 1) t += (uint64_t)m[i] * P::MOD[k - i];
            else u += (uint64_t)m[i] * P::MOD[k - i];
        }
        t += u;
        u = 0;
        m[k] = ((uint32_t)t * P::NINV) & M29;
        t += (uint64_t)m[k] * P::MOD[0];
        t >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
        col(k, t, u);
#pragma unroll
        for (int i = k - (NL - 1); i < NL; ++i) {
            if (i & 1) t += (uint64_t)m[i] * P::MOD[k - i];
            else u += (uint64_t)m[i] * P::MOD[k - i];
        }
        t += u;
        u = 0;
        r.l[k - NL] = (uint32_t)t & M29;
        t >>= 29;
    }
    r.l[NL - 1] = (uint32_t)t;
    return r;
}
template <class P>
FP_INLINE void fp_col_mul(int k, const Fp<P>& a, const Fp<P>& b, uint64_t& t, uint64_t& u) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int j = k - i;
        if (j >= 0 && j < NL) {
            if (i & 1) u += (uint64_t)a.l[i] * b.l[j];
            else t += (uint64_t)a.l[i] * b.l[j];
        }
    }
}
template <class P>
FP_INLINE Fp<P> fp_mul_ps(const Fp<P>& a, const Fp<P>& b) {
    return fp_mont_ps<P>([&](int k, uint64_t& t, uint64_t& u) { fp_col_mul<P>(k, a, b, t, u); });
}
template <class P>
FP_INLINE Fp<P> fp_mul2_ps(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    return fp_mont_ps<P>([&](int k, uint64_t& t, uint64_t& u) {
        fp_col_mul<P>(k, a, b, t, u);
        fp_col_mul<P>(k, c, d, u, t);
    });
}
// (a*b + c*d + e*f) / 2^261 mod m with ONE reduction: 27 + 9 products of < 2^58 per column still fit 64 bits.
// Output bound: (A*B + C*D + E*F)/169 + 1.
template <class P>
FP_INLINE Fp<P> fp_mul3_ps(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, const Fp<P>& e, const Fp<P>& f) {
    return fp_mont_ps<P>([&](int k, uint64_t& t, uint64_t& u) {
        fp_col_mul<P>(k, a, b, t, u);
        fp_col_mul<P>(k, c, d, u, t);
        fp_col_mul<P>(k, e, f, t, u);
    });
}
template <class P>
FP_INLINE Fp<P> fp_sqr_ps(const Fp<P>& a) {
    return fp_mont_ps<P>([&](int k, uint64_t& t, uint64_t& u) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int j = k - i;
            if (j > i && j < NL) {  // cross terms once, with a doubled operand (< 2^30)
                if (i & 1) u += (uint64_t)(a.l[i] << 1) * a.l[j];
                else t += (uint64_t)(a.l[i] << 1) * a.l[j];
            }
        }
        if ((k & 1) == 0) t += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
    });
}

// ---- hand-scheduled multiply-add chains ------------------------------------------------------------------------
// Written as C++ (`t += (uint64_t)a * b`), a column of the product comes out of LLVM as a fresh chain that starts from 0,
// merged into the running carry by a 64-bit add (v_lshl_add_u64, half rate like the multiply-add itself): the reassociation
// pass does that to every formulation, one or two partial sums alike — 16-17 merges per product, 7 % of a mixed addition.
// The multiply-add instruction takes its 64-bit addend for free, so a column can start FROM the carry; that makes the whole
// product one dependent chain, and the parallelism the merges bought has to come from somewhere else: two independent
// products side by side (the group law offers them in pairs), written alternately.  An empty asm on the running sum after every
// multiply-add keeps LLVM from reassociating (a real `v_mad_u64_u32` asm works too, but the hazard recogniser then puts an
// s_nop behind every one of them).
FP_INLINE void mad_vv(uint64_t& t, uint32_t a, uint32_t b) {
    t += (uint64_t)a * b;
    asm("" : "+v"(t));   // no instruction: keeps this multiply-add's sum out of the reassociation of the column
}
FP_INLINE void mad_vs(uint64_t& t, uint32_t a, uint32_t k) {   // k: wave-uniform (a modulus limb)
    t += (uint64_t)a * k;
    asm("" : "+v"(t));
}
FP_INLINE uint64_t mul_vv(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    asm("" : "+v"(t));
    return t;
}
// Two Montgomery products in lock step.  term(k, i, x0, y0, x1, y1) yields the i-th product term of column k of both
// products (or returns false when column k has no i-th term).
template <class P, class TermF>
FP_INLINE void fp_mont_chain2(TermF&& term, Fp<P>& r0, Fp<P>& r1) {
    uint32_t m0[NL], m1[NL];
    uint64_t t0 = 0, t1 = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
        bool first = (k == 0);
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i) {
            uint32_t x0, y0, x1, y1;
            if (term(k, i, x0, y0, x1, y1)) {
                if (first) {
                    t0 = mul_vv(x0, y0);
                    t1 = mul_vv(x1, y1);
                    first = false;
                } else {
                    mad_vv(t0, x0, y0);
                    mad_vv(t1, x1, y1);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int j = k - i;
            if (i < k && j >= 0 && j < NL) {   // m_i * p_(k-i), i < k
                mad_vs(t0, m0[i], P::MOD[j]);
                mad_vs(t1, m1[i], P::MOD[j]);
            }
        }
        if (k < NL) {
            m0[k] = ((uint32_t)t0 * P::NINV) & M29;
            m1[k] = ((uint32_t)t1 * P::NINV) & M29;
            mad_vs(t0, m0[k], P::MOD[0]);
            mad_vs(t1, m1[k], P::MOD[0]);
        } else {
            r0.l[k - NL] = (uint32_t)t0 & M29;
            r1.l[k - NL] = (uint32_t)t1 & M29;
        }
        t0 >>= 29;
        t1 >>= 29;
    }
    r0.l[NL - 1] = (uint32_t)t0;
    r1.l[NL - 1] = (uint32_t)t1;
}
// (r0, r1) = (a*b, c*d)
template <class P>
FP_INLINE void fp_mul_dual(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, Fp<P>& r0, Fp<P>& r1) {
    fp_mont_chain2<P>([&](int k, int i, uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1) {
        const int j = k - i;
        if (i >= NL || j < 0 || j >= NL) return false;
        x0 = a.l[i]; y0 = b.l[j]; x1 = c.l[i]; y1 = d.l[j];
        return true;
    }, r0, r1);
}
// (r0, r1) = (a^2, c^2): cross terms once with a doubled operand
template <class P>
FP_INLINE void fp_sqr_dual(const Fp<P>& a, const Fp<P>& c, Fp<P>& r0, Fp<P>& r1) {
    uint32_t a2[NL], c2[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        a2[i] = a.l[i] << 1;
        c2[i] = c.l[i] << 1;
    }
    fp_mont_chain2<P>([&](int k, int i, uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1) {
        const int j = k - i;
        if (i >= NL || j < i || j >= NL) return false;
        if (j == i) { x0 = a.l[i]; y0 = a.l[i]; x1 = c.l[i]; y1 = c.l[i]; }
        else { x0 = a2[i]; y0 = a.l[j]; x1 = c2[i]; y1 = c.l[j]; }
        return true;
    }, r0, r1);
}
// (r0, r1, r2) = (a*b + c*d, e*f, g*h): a two-product sum (one reduction) beside two plain products, three chains
template <class P>
FP_INLINE void fp_mul2_mul_mul(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, const Fp<P>& e,
                               const Fp<P>& f, const Fp<P>& g, const Fp<P>& h, Fp<P>& r0, Fp<P>& r1, Fp<P>& r2) {
    uint32_t m0[NL], m1[NL], m2[NL];
    uint64_t t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int j = k - i;
            if (j >= 0 && j < NL) {
                if (k == 0) {
                    t0 = mul_vv(a.l[i], b.l[j]);
                    t1 = mul_vv(e.l[i], f.l[j]);
                    t2 = mul_vv(g.l[i], h.l[j]);
                } else {
                    mad_vv(t0, a.l[i], b.l[j]);
                    mad_vv(t1, e.l[i], f.l[j]);
                    mad_vv(t2, g.l[i], h.l[j]);
                }
                mad_vv(t0, c.l[i], d.l[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int j = k - i;
            if (i < k && j >= 0 && j < NL) {
                mad_vs(t0, m0[i], P::MOD[j]);
                mad_vs(t1, m1[i], P::MOD[j]);
                mad_vs(t2, m2[i], P::MOD[j]);
            }
        }
        if (k < NL) {
            m0[k] = ((uint32_t)t0 * P::NINV) & M29;
            m1[k] = ((uint32_t)t1 * P::NINV) & M29;
            m2[k] = ((uint32_t)t2 * P::NINV) & M29;
            mad_vs(t0, m0[k], P::MOD[0]);
            mad_vs(t1, m1[k], P::MOD[0]);
            mad_vs(t2, m2[k], P::MOD[0]);
        } else {
            r0.l[k - NL] = (uint32_t)t0 & M29;
            r1.l[k - NL] = (uint32_t)t1 & M29;
            r2.l[k - NL] = (uint32_t)t2 & M29;
        }
        t0 >>= 29;
        t1 >>= 29;
        t2 >>= 29;
    }
    r0.l[NL - 1] = (uint32_t)t0;
    r1.l[NL - 1] = (uint32_t)t1;
    r2.l[NL - 1] = (uint32_t)t2;
}

// a*b / 2^261 mod m.  Inputs: tight limbs, bounds A, B with A*B <= ~1000.  Output bound: A*B/169 + 1.
template <class P>
FP_INLINE Fp<P> fp_mul_os(const Fp<P>& a, const Fp<P>& b) {
    uint64_t acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
    return fp_mont_reduce<P>(acc);
}

// (a*b + c*d) / 2^261 mod m with ONE Montgomery reduction (27 products of < 2^58 per column still fit
// 64 bits).  Output bound: (A*B + C*D)/169 + 1.
template <class P>
FP_INLINE Fp<P> fp_mul2_os(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    uint64_t acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[i + j] += (uint64_t)c.l[i] * d.l[j];
    }
    return fp_mont_reduce<P>(acc);
}

// a - b - 2c + K*m in one carry sweep.  REQUIRES value(b) + 2*value(c) <= K*m.  bound: A + K.
template <int K, class P>
FP_INLINE Fp<P> fp_sub_sub2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i)
        x[i] = (int32_t)(a.l[i] + km_limb<P>(K, i)) - (int32_t)b.l[i] - (int32_t)(c.l[i] << 1);
    return fp_normalize<P>(x);
}
// a - 2b + K*m.  REQUIRES 2*value(b) <= K*m.  bound: A + K.
template <int K, class P>
FP_INLINE Fp<P> fp_sub2(const Fp<P>& a, const Fp<P>& b) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)(a.l[i] + km_limb<P>(K, i)) - (int32_t)(b.l[i] << 1);
    return fp_normalize<P>(x);
}
// 3a.  bound: 3A.
template <class P>
FP_INLINE Fp<P> fp_triple(const Fp<P>& a) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)(a.l[i] * 3u);
    return fp_normalize<P>(x);
}

// a*a / 2^261 mod m: 45 products instead of 81.  Bound as fp_mul.
template <class P>
FP_INLINE Fp<P> fp_sqr_os(const Fp<P>& a) {
    uint64_t acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0;
    uint32_t d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = a.l[i] << 1;  // < 2^30
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        acc[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < NL; ++j) acc[i + j] += (uint64_t)d[i] * a.l[j];
    }
    return fp_mont_reduce<P>(acc);
}

// The forms used everywhere (H2AGG_FP_OPERAND_SCANNING selects the 18-column variant for A/B measurements)
#ifdef H2AGG_FP_OPERAND_SCANNING
template <class P> FP_INLINE Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) { return fp_mul_os<P>(a, b); }
template <class P> FP_INLINE Fp<P> fp_sqr(const Fp<P>& a) { return fp_sqr_os<P>(a); }
template <class P> FP_INLINE Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) { return fp_mul2_os<P>(a, b, c, d); }
#else
template <class P> FP_INLINE Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) { return fp_mul_ps<P>(a, b); }
template <class P> FP_INLINE Fp<P> fp_sqr(const Fp<P>& a) { return fp_sqr_ps<P>(a); }
template <class P> FP_INLINE Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) { return fp_mul2_ps<P>(a, b, c, d); }
#endif

// value < 2m  ->  [0, m)
template <class P>
FP_INLINE Fp<P> fp_cond_sub(const Fp<P>& a) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)a.l[i] - (int32_t)km_limb<P>(1, i);
    Fp<P> t = fp_normalize<P>(x);
    const bool neg = (int32_t)t.l[8] < 0;
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = neg ? a.l[i] : t.l[i];
    return r;
}

// any value < 2^261  ->  the canonical representative in [0, m)   (one multiplication by R mod m)
template <class P>
FP_INLINE Fp<P> fp_canonical(const Fp<P>& a) {
    return fp_cond_sub<P>(fp_mul<P>(a, Fp<P>::one()));
}

// Could a (value < K*m) be a multiple of m?  Exact "no", cheap "maybe": compares the low limb with the
// low limbs of 0, m, ..., (K-1)*m.
template <int K, class P>
FP_INLINE bool fp_maybe_zero_mod(const Fp<P>& a) {
    // a = k*m with 0 <= k < K  =>  a.l[0] = k*m_0 mod 2^29  =>  k = a.l[0] * m_0^-1 mod 2^29, and
    // m_0^-1 = 2^29 - NINV.  One multiply instead of K compares; never misses a multiple.
    const uint32_t k = (a.l[0] * ((1u << 29) - P::NINV)) & M29;
    return k < (uint32_t)K;
}
// The same filter on TWO limbs (2^-58 instead of 2^-29 per candidate): for callers whose "maybe" is expensive — the lean
// accumulation hands every "maybe" to a separate fix-up pass.  Still never misses a multiple.  a: tight limbs.
template <int K, class P>
FP_INLINE bool fp_maybe_zero_mod2(const Fp<P>& a) {
    const uint32_t k = (a.l[0] * ((1u << 29) - P::NINV)) & M29;
    if (k >= (uint32_t)K) return false;
    const uint64_t c = (uint64_t)k * P::MOD[0];   // limb 1 of k * m
    const uint32_t l1 = (uint32_t)((c >> 29) + (uint64_t)k * P::MOD[1]) & M29;
    return a.l[1] == l1;
}
// exact test a == 0 (mod m), value(a) < K*m
template <int K, class P>
FP_INLINE bool fp_is_zero_mod(const Fp<P>& a) {
    if (!fp_maybe_zero_mod<K, P>(a)) return false;
    return fp_canonical<P>(a).is_zero_int();
}

// canonical integer (tight limbs, < m) -> Montgomery (< 2m)
template <class P>
FP_INLINE Fp<P> fp_to_mont(const Fp<P>& a) {
    return fp_mul<P>(a, Fp<P>::r2());
}
// Montgomery (any bound) -> canonical integer in [0, m)
template <class P>
FP_INLINE Fp<P> fp_from_mont(const Fp<P>& a) {
    Fp<P> one;
#pragma unroll
    for (int i = 0; i < NL; ++i) one.l[i] = (i == 0);
    return fp_cond_sub<P>(fp_mul<P>(a, one));
}

// ---- modular inverse: safegcd (Bernstein-Yang divsteps, "half-delta" variant) on signed 29-bit limbs -----------
// 21 rounds of 29 division steps (609 >= the 590 that 256-bit inputs need), each round: a 2x2 transition matrix from
// the low limbs of (f, g) with branch-free integer ops, then (f, g) <- M (f, g) / 2^29 exactly and
// (d, e) <- M (d, e) / 2^29 mod m.  ~13 k instructions against ~85 k for the Fermat ladder (254 squarings + ~127
// multiplications) it replaces: a to_affine is one inversion of pure dependent latency on its lane, 0.2 ms before.
// Invariants: d * x = f, e * x = g (mod m, up to the powers of two divided out); f ends as +-1 with g = 0.
struct Trans2x2 {
    int32_t u, v, q, r;
};
FP_INLINE int32_t fp_divsteps29(int32_t zeta, uint32_t f0, uint32_t g0, Trans2x2& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 29; ++i) {
        uint32_t m1 = (uint32_t)(zeta >> 31);            // zeta < 0  (delta > 0)
        const uint32_t m2 = 0u - (g & 1u);               // g odd
        const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;   // -f, -u, -v when zeta < 0
        g += x & m2;
        q += y & m2;
        r += z & m2;
        m1 &= m2;
        zeta = (int32_t)(((uint32_t)zeta ^ m1) - 1u);     // -zeta - 2 on a swap, zeta - 1 otherwise
        f += g & m1;
        u += q & m1;
        v += r & m1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}
// x^-1 mod m for a canonical integer x (tight limbs, < m); canonical out; 0 -> 0
template <class P>
__device__ __noinline__ Fp<P> fp_inv_int(const Fp<P>& x) {
    constexpr int32_t MASK = (int32_t)M29;
    constexpr uint32_t MINV = ((1u << 29) - P::NINV) & M29;   // m^-1 mod 2^29
    int32_t f[NL], g[NL], d[NL], e[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        f[i] = (int32_t)P::MOD[i];
        g[i] = (int32_t)x.l[i];
        d[i] = 0;
        e[i] = (i == 0);
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 21; ++it) {
        Trans2x2 t;
        zeta = fp_divsteps29(zeta, (uint32_t)f[0] | ((uint32_t)f[1] << 29), (uint32_t)g[0] | ((uint32_t)g[1] << 29), t);
        {   // (d, e) <- t * (d, e) / 2^29 mod m, both kept in (-2m, m)
            const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
            int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
            int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0], ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
            md -= (int32_t)((MINV * (uint32_t)cd + (uint32_t)md) & M29);   // make the low 29 bits of cd + m * md vanish
            me -= (int32_t)((MINV * (uint32_t)ce + (uint32_t)me) & M29);
            cd += (int64_t)(int32_t)P::MOD[0] * md;
            ce += (int64_t)(int32_t)P::MOD[0] * me;
            cd >>= 29;
            ce >>= 29;
#pragma unroll
            for (int i = 1; i < NL; ++i) {
                cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)(int32_t)P::MOD[i] * md;
                ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)(int32_t)P::MOD[i] * me;
                d[i - 1] = (int32_t)cd & MASK;
                e[i - 1] = (int32_t)ce & MASK;
                cd >>= 29;
                ce >>= 29;
            }
            d[NL - 1] = (int32_t)cd;
            e[NL - 1] = (int32_t)ce;
        }
        {   // (f, g) <- t * (f, g) / 2^29, exact
            int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0], cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
            cf >>= 29;
            cg >>= 29;
#pragma unroll
            for (int i = 1; i < NL; ++i) {
                cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
                cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
                f[i - 1] = (int32_t)cf & MASK;
                g[i - 1] = (int32_t)cg & MASK;
                cf >>= 29;
                cg >>= 29;
            }
            f[NL - 1] = (int32_t)cf;
            g[NL - 1] = (int32_t)cg;
        }
    }
    // d is in (-2m, m) and f = +-1: add m if negative, negate if f = -1, add m again if still negative
    int32_t ca = d[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] += (int32_t)P::MOD[i] & ca;
    const int32_t cn = f[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (d[i] ^ cn) - cn;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        d[i + 1] += d[i] >> 29;
        d[i] &= MASK;
    }
    ca = d[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] += (int32_t)P::MOD[i] & ca;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        d[i + 1] += d[i] >> 29;
        d[i] &= MASK;
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = (uint32_t)d[i];
    return r;
}
// Montgomery in (any bound) / out (< 2m).  inv(0) = 0; callers mirroring `invert().unwrap()` test first.
template <class P>
FP_INLINE Fp<P> fp_inv(const Fp<P>& a) {
    return fp_to_mont<P>(fp_inv_int<P>(fp_from_mont<P>(a)));   // (x R) -> x -> x^-1 -> x^-1 R
}

// ---- 32-byte packed form (8 x 32-bit words, value < 2^256) <-> limbs -------------------------------
template <class P>
FP_INLINE Fp<P> fp_unpack(const uint32_t (&w)[8]) {
    Fp<P> r;
    r.l[0] = w[0] & M29;
    r.l[1] = ((w[0] >> 29) | (w[1] << 3)) & M29;
    r.l[2] = ((w[1] >> 26) | (w[2] << 6)) & M29;
    r.l[3] = ((w[2] >> 23) | (w[3] << 9)) & M29;
    r.l[4] = ((w[3] >> 20) | (w[4] << 12)) & M29;
    r.l[5] = ((w[4] >> 17) | (w[5] << 15)) & M29;
    r.l[6] = ((w[5] >> 14) | (w[6] << 18)) & M29;
    r.l[7] = ((w[6] >> 11) | (w[7] << 21)) & M29;
    r.l[8] = w[7] >> 8;
    return r;
}
template <class P>
FP_INLINE void fp_pack(uint32_t (&w)[8], const Fp<P>& a) {  // value(a) < 2^256, tight limbs
    w[0] = a.l[0] | (a.l[1] << 29);
    w[1] = (a.l[1] >> 3) | (a.l[2] << 26);
    w[2] = (a.l[2] >> 6) | (a.l[3] << 23);
    w[3] = (a.l[3] >> 9) | (a.l[4] << 20);
    w[4] = (a.l[4] >> 12) | (a.l[5] << 17);
    w[5] = (a.l[5] >> 15) | (a.l[6] << 14);
    w[6] = (a.l[6] >> 18) | (a.l[7] << 11);
    w[7] = (a.l[7] >> 21) | (a.l[8] << 8);
}
// 32 bytes in global memory (two 16-byte accesses) <-> limbs
template <class P>
FP_INLINE Fp<P> fp_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fp_unpack<P>(w);
}
template <class P>
FP_INLINE void fp_store(void* p, const Fp<P>& v) {
    uint32_t w[8];
    fp_pack<P>(w, v);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// is the 256-bit integer at p (as loaded by fp_load) < m ?  (top limb of a 256-bit load is < 2^24)
template <class P>
FP_INLINE bool fp_is_canonical(const Fp<P>& a) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)a.l[i] - (int32_t)km_limb<P>(1, i);
    Fp<P> t = fp_normalize<P>(x);
    return (int32_t)t.l[8] < 0;
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// Montgomery products as single inline-asm blocks with fixed temporaries (v108..v127): for kernels that must fit 128 VGPRs
#include "fp_asm.inc"

}  // namespace h2agg
