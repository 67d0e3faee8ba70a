// BN254 prime-field arithmetic for gfx950 (CDNA4), device side.
//
// Stands behind the halo2curves `bn256::{Fq,Fr}` operators the reference calls from
//   halo2-snark-aggregator-api/src/mock/arith/field.rs:45,54,104,113,121,132,144   (Fr: add/sub/mul/invert)
//   halo2-snark-aggregator-api/src/mock/arith/ecc.rs:36,45,94,103                  (Fq inside every G1 op)
//
// Representation: 8 x 32-bit little-endian limbs in VGPRs, Montgomery form with R = 2^256, always
// fully reduced to [0, m) so equality / zero tests (needed for the exceptional cases of the group law —
// results must be bit-exact) are plain limb compares.  Multiplication is CIOS built on
// v_mad_u64_u32 (32x32+64 -> 64); both moduli are < 2^254 so the running value never needs a ninth
// limb.  No MFMA: this is carry-chained integer arithmetic, not a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FP_INLINE __device__ __forceinline__

namespace h2agg {

struct FqParams {
    // p = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xe4866389u;  // -p^-1 mod 2^32
    static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                       0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrParams {
    // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
    static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                       0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class P>
struct Fp {
    uint32_t l[8];

    static FP_INLINE Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = 0;
        return r;
    }
    static FP_INLINE Fp one() {  // Montgomery 1
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = P::R1[i];
        return r;
    }
    static FP_INLINE Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = P::R2[i];
        return r;
    }
    FP_INLINE bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i];
        return o == 0;
    }
    FP_INLINE bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= (l[i] ^ b.l[i]);
        return o == 0;
    }
    FP_INLINE bool operator!=(const Fp& b) const { return !(*this == b); }
};

// t = a - m ; returns borrow (1 if a < m)
template <class P>
FP_INLINE uint32_t sub_mod_raw(uint32_t (&t)[8], const uint32_t (&a)[8]) {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a[i] - P::MOD[i] - br;
        t[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    return (uint32_t)br;
}

// conditional final subtraction: a in [0, 2m) -> [0, m)
template <class P>
FP_INLINE void reduce_once(uint32_t (&a)[8]) {
    uint32_t t[8];
    uint32_t borrow = sub_mod_raw<P>(t, a);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = borrow ? a[i] : t[i];
}

template <class P>
FP_INLINE Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    // a + b < 2m < 2^255: no carry out of limb 7
    reduce_once<P>(r.l);
    return r;
}

template <class P>
FP_INLINE Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
        r.l[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)r.l[i] + (P::MOD[i] & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
FP_INLINE Fp<P> fp_neg(const Fp<P>& a) {
    Fp<P> r;
    uint64_t br = 0;
    bool z = a.is_zero();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)P::MOD[i] - a.l[i] - br;
        r.l[i] = z ? 0u : (uint32_t)d;
        br = (d >> 32) & 1;
    }
    return r;
}

template <class P>
FP_INLINE Fp<P> fp_dbl(const Fp<P>& a) {
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.l[i] = (a.l[i] << 1) | c;
        c = a.l[i] >> 31;
    }
    reduce_once<P>(r.l);
    return r;
}

// Montgomery product a*b*2^-256 mod m.  CIOS, 8x8 limbs, 2 x 64 v_mad_u64_u32.
template <class P>
FP_INLINE Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.l[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        const uint32_t t8 = (uint32_t)c;
        const uint32_t m = t[0] * P::INV;
        c = (uint64_t)m * P::MOD[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)m * P::MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t8;
        t[7] = (uint32_t)c;  // value stays < 2m < 2^255: (c >> 32) == 0
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[i];
    reduce_once<P>(r.l);
    return r;
}

template <class P>
FP_INLINE Fp<P> fp_sqr(const Fp<P>& a) {
    return fp_mul<P>(a, a);
}

// canonical integer (8 LE limbs, < m) -> Montgomery
template <class P>
FP_INLINE Fp<P> fp_to_mont(const Fp<P>& a) {
    return fp_mul<P>(a, Fp<P>::r2());
}
// Montgomery -> canonical integer
template <class P>
FP_INLINE Fp<P> fp_from_mont(const Fp<P>& a) {
    Fp<P> one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.l[i] = (i == 0);
    return fp_mul<P>(a, one);
}

// a^(m-2) (Fermat).  inv(0) = 0; callers that mirror `invert().unwrap()` test for zero first.
template <class P>
__device__ __noinline__ Fp<P> fp_inv(const Fp<P>& a) {
    Fp<P> acc = Fp<P>::one();
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        // i is a compile-time constant after unrolling, so MOD[i] folds to a literal
        const uint32_t e = P::MOD[i] - (i == 0 ? 2u : 0u);  // low limbs of both moduli are >= 2: no borrow
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            acc = fp_sqr<P>(acc);
            if ((e >> bit) & 1) acc = fp_mul<P>(acc, a);
        }
    }
    return acc;
}

// 32-byte global <-> registers (two 16-byte accesses per element)
template <class P>
FP_INLINE Fp<P> fp_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fp<P> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class P>
FP_INLINE void fp_store(void* p, const Fp<P>& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// is the canonical integer < m ?
template <class P>
FP_INLINE bool fp_is_canonical(const Fp<P>& a) {
    uint32_t t[8];
    return sub_mod_raw<P>(t, a.l) != 0;
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

}  // namespace h2agg
