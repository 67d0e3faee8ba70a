// Batched Fr / G1 kernels (HBM-shaped: one element per lane, 16-byte vector loads, grid-stride).
// Each kernel names the reference operation it stands behind.
#pragma once
#include "g1.hpp"

namespace h2agg {

// status bits a kernel may raise (atomicOr into ctx->d_flags)
enum : uint32_t { FLAG_NONCANONICAL = 1u, FLAG_DIV_ZERO = 2u, FLAG_BAD_POINT = 4u };

constexpr int BLOCK = 256;

// Start of a synchronous entry point: flags[0] (this call's status) is rolled into flags[2] (sticky: raised by earlier
// ASYNCHRONOUS calls that nobody has reported yet) instead of being wiped.
__global__ void k_flags_roll(uint32_t* flags) {
    flags[2] |= flags[0];
    flags[0] = 0;
}

// ------------------------------------------------------------------ block-wide reductions through LDS
// SoA layout lds[k * BLOCK + tid] (k = limb index) keeps every ds access conflict-free.
template <int NLIMB>
FP_INLINE void lds_put(uint32_t* lds, int tid, const uint32_t* v) {
#pragma unroll
    for (int k = 0; k < NLIMB; ++k) lds[k * BLOCK + tid] = v[k];
}
template <int NLIMB>
FP_INLINE void lds_get(const uint32_t* lds, int tid, uint32_t* v) {
#pragma unroll
    for (int k = 0; k < NLIMB; ++k) v[k] = lds[k * BLOCK + tid];
}
constexpr int XYZZ_WORDS = 4 * NL;  // 36
FP_INLINE void lds_put_xyzz(uint32_t* lds, int tid, const G1XYZZ& p) {
    lds_put<NL>(lds, tid, p.x.l);
    lds_put<NL>(lds + NL * BLOCK, tid, p.y.l);
    lds_put<NL>(lds + 2 * NL * BLOCK, tid, p.zz.l);
    lds_put<NL>(lds + 3 * NL * BLOCK, tid, p.zzz.l);
}
FP_INLINE G1XYZZ lds_get_xyzz(const uint32_t* lds, int tid) {
    G1XYZZ p;
    lds_get<NL>(lds, tid, p.x.l);
    lds_get<NL>(lds + NL * BLOCK, tid, p.y.l);
    lds_get<NL>(lds + 2 * NL * BLOCK, tid, p.zz.l);
    lds_get<NL>(lds + 3 * NL * BLOCK, tid, p.zzz.l);
    return p;
}
// Sum of one XYZZ point per thread over a BLOCK-thread workgroup; result valid in thread 0.
// lds must hold XYZZ_WORDS * BLOCK words.
__device__ __noinline__ G1XYZZ block_sum_xyzz(G1XYZZ v, uint32_t* lds) {
    const int tid = threadIdx.x;
    lds_put_xyzz(lds, tid, v);
    __syncthreads();
#pragma unroll 1
    for (int s = BLOCK / 2; s >= 1; s >>= 1) {
        if (tid < s) {
            G1XYZZ o = lds_get_xyzz(lds, tid + s);
            v = xyzz_add(v, o);
            lds_put_xyzz(lds, tid, v);
        }
        __syncthreads();
    }
    return v;
}
// canonical (< r) in, canonical out; lds must hold NL * BLOCK words
__device__ __noinline__ Fr block_sum_fr(Fr v, uint32_t* lds) {
    const int tid = threadIdx.x;
    lds_put<NL>(lds, tid, v.l);
    __syncthreads();
#pragma unroll 1
    for (int s = BLOCK / 2; s >= 1; s >>= 1) {
        if (tid < s) {
            Fr o;
            lds_get<NL>(lds, tid + s, o.l);
            v = fp_cond_sub<FrParams>(fp_add<FrParams>(v, o));
            lds_put<NL>(lds, tid, v.l);
        }
        __syncthreads();
    }
    return v;
}

// ------------------------------------------------------------------ Fr element-wise
// MockFieldChip::{add,sub,mul,square,div(->inverse)}  mock/arith/field.rs:39-55, 98-122
__global__ void __launch_bounds__(BLOCK) k_fr_batch_op(int op, const uint8_t* __restrict__ a,
                                                       const uint8_t* __restrict__ b, size_t n,
                                                       uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Fr x = fp_load<FrParams>(a + 32 * i);
        uint32_t bad = !fp_is_canonical<FrParams>(x);
        Fr y = Fr::zero();
        if (op <= 2 || op == 5) {
            y = fp_load<FrParams>(b + 32 * i);
            bad |= !fp_is_canonical<FrParams>(y);
        }
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        Fr z;  // canonical inputs (< r); every branch ends in the canonical representative
        switch (op) {
        case 0: z = fp_cond_sub<FrParams>(fp_add<FrParams>(x, y)); break;
        case 1: z = fp_cond_sub<FrParams>(fp_sub<1, FrParams>(x, y)); break;
        case 2: z = fp_cond_sub<FrParams>(fp_mul<FrParams>(fp_to_mont<FrParams>(x), y)); break;  // (xR)*y/R = xy
        case 3: z = fp_cond_sub<FrParams>(fp_mul<FrParams>(fp_to_mont<FrParams>(x), x)); break;
        case 5:   // div: a * b.invert().unwrap()   (mock/arith/field.rs:107-114)
            if (y.is_zero_int()) atomicOr(flags, FLAG_DIV_ZERO);
            z = fp_cond_sub<FrParams>(fp_mul<FrParams>(fp_inv<FrParams>(fp_to_mont<FrParams>(y)), x));  // (R/y) * x / R
            break;
        default:
            if (x.is_zero_int()) atomicOr(flags, FLAG_DIV_ZERO);
            z = fp_from_mont<FrParams>(fp_inv<FrParams>(fp_to_mont<FrParams>(x)));
            break;
        }
        fp_store<FrParams>(out + 32 * i, z);
    }
}

// x^e for a small runtime exponent (Montgomery in/out)
__device__ __noinline__ Fr fr_pow_u64(Fr x, uint64_t e) {
    Fr acc = Fr::one();
#pragma unroll 1
    for (int bit = 63; bit >= 0; --bit) {
        acc = fp_sqr<FrParams>(acc);
        if ((e >> bit) & 1) acc = fp_mul<FrParams>(acc, x);
    }
    return acc;
}

// ArithFieldChip::pow_constant (arith/field.rs:83-104): out_i = a_i^e for one exponent e >= 1
__global__ void __launch_bounds__(BLOCK) k_fr_batch_pow(const uint8_t* __restrict__ a, size_t n, uint64_t e,
                                                        uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Fr x = fp_load<FrParams>(a + 32 * i);
        if (!fp_is_canonical<FrParams>(x)) atomicOr(flags, FLAG_NONCANONICAL);
        fp_store<FrParams>(out + 32 * i, fp_from_mont<FrParams>(fr_pow_u64(fp_to_mont<FrParams>(x), e)));
    }
}

// ArithFieldChip::mul_add_accumulate default: acc = acc*b + v_i  (arith/field.rs:68-81)
// Horner over n terms = sum v_i b^(n-1-i).  One workgroup: the sequence is left-padded with zeros to
// BLOCK chunks of L terms, thread t Horner-evaluates chunk t, then thread 0 Horner-combines the BLOCK
// partials with base b^L.
__global__ void __launch_bounds__(BLOCK) k_fr_horner(const uint8_t* __restrict__ v, size_t n,
                                                     const uint8_t* __restrict__ b_in, uint8_t* __restrict__ out,
                                                     uint32_t* flags) {
    __shared__ uint32_t lds[NL * BLOCK];
    const int tid = threadIdx.x;
    const size_t L = (n + BLOCK - 1) / BLOCK;
    const size_t pad = L * BLOCK - n;
    Fr b = fp_load<FrParams>(b_in);
    uint32_t bad = !fp_is_canonical<FrParams>(b);
    b = fp_to_mont<FrParams>(b);
    Fr acc = Fr::zero();
#pragma unroll 1
    for (size_t k = 0; k < L; ++k) {
        size_t g = (size_t)tid * L + k;  // index in the padded sequence
        Fr x = Fr::zero();
        if (g >= pad) {
            x = fp_load<FrParams>(v + 32 * (g - pad));
            bad |= !fp_is_canonical<FrParams>(x);
            x = fp_to_mont<FrParams>(x);
        }
        acc = fp_add<FrParams>(fp_mul<FrParams>(acc, b), x);
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    lds_put<NL>(lds, tid, acc.l);
    __syncthreads();
    if (tid == 0) {
        Fr bl = fr_pow_u64(b, (uint64_t)L);
        Fr r = Fr::zero();
#pragma unroll 1
        for (int t = 0; t < BLOCK; ++t) {
            Fr h;
            lds_get<NL>(lds, t, h.l);
            r = fp_add<FrParams>(fp_mul<FrParams>(r, bl), h);
        }
        fp_store<FrParams>(out, fp_from_mont<FrParams>(r));
    }
}

// MockFieldChip::sum_with_coeff_and_constant: b + sum x_i*coeff_i  (mock/arith/field.rs:124-135)
__global__ void __launch_bounds__(BLOCK) k_fr_sum_coeff(const uint8_t* __restrict__ x, const uint8_t* __restrict__ cf,
                                                        size_t n, const uint8_t* __restrict__ b_in,
                                                        uint8_t* __restrict__ out, uint32_t* flags) {
    __shared__ uint32_t lds[NL * BLOCK];
    Fr acc = Fr::zero();  // canonical-domain accumulator: (xR)*c/R = x*c canonical
    uint32_t bad = 0;
    for (size_t i = threadIdx.x; i < n; i += BLOCK) {
        Fr a = fp_load<FrParams>(x + 32 * i);
        Fr c = fp_load<FrParams>(cf + 32 * i);
        bad |= !fp_is_canonical<FrParams>(a) | !fp_is_canonical<FrParams>(c);
        Fr t = fp_cond_sub<FrParams>(fp_mul<FrParams>(fp_to_mont<FrParams>(a), c));
        acc = fp_cond_sub<FrParams>(fp_add<FrParams>(acc, t));
    }
    Fr tot = block_sum_fr(acc, lds);
    if (threadIdx.x == 0) {
        Fr b = fp_load<FrParams>(b_in);
        bad |= !fp_is_canonical<FrParams>(b);
        fp_store<FrParams>(out, fp_cond_sub<FrParams>(fp_add<FrParams>(tot, b)));
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
}

// ------------------------------------------------------------------ G1 element-wise
FP_INLINE uint32_t jac_noncanonical(const uint8_t* p) {
    return !fp_is_canonical<FqParams>(fp_load<FqParams>(p)) | !fp_is_canonical<FqParams>(fp_load<FqParams>(p + 32)) |
           !fp_is_canonical<FqParams>(fp_load<FqParams>(p + 64));
}

// MockEccChip::add / sub  (mock/arith/ecc.rs:30-46)
__global__ void __launch_bounds__(BLOCK) k_g1_batch_add(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                        size_t n, int subtract, uint8_t* __restrict__ out,
                                                        uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        if (jac_noncanonical(a + 96 * i) | jac_noncanonical(b + 96 * i)) atomicOr(flags, FLAG_NONCANONICAL);
        G1XYZZ p = xyzz_from_jac(jac_load_canonical(a + 96 * i));
        G1XYZZ q = xyzz_from_jac(jac_load_canonical(b + 96 * i));
        if (subtract) q = xyzz_neg(q);
        jac_store_canonical(out + 96 * i, jac_from_xyzz(xyzz_add(p, q)));
    }
}

// canonical affine (64 B) -> Montgomery affine; identity (0,0) stays (0,0)
FP_INLINE G1Affine affine_load_canonical(const uint8_t* p, uint32_t& bad) {
    G1Affine r;
    Fq x = fp_load<FqParams>(p), y = fp_load<FqParams>(p + 32);
    bad |= !fp_is_canonical<FqParams>(x) | !fp_is_canonical<FqParams>(y);
    r.x = fp_to_mont<FqParams>(x);
    r.y = fp_to_mont<FqParams>(y);
    return r;
}

// a canonical 256-bit scalar as eight 32-bit words (what the C ABI carries; digits are taken from it directly)
struct U256 {
    uint32_t w[8];
};
FP_INLINE U256 u256_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    U256 r;
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
    r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
    return r;
}
FP_INLINE bool u256_is_canonical_fr(const U256& s) { return fp_is_canonical<FrParams>(fp_unpack<FrParams>(s.w)); }

// s * P, MSB-first double-and-add on the canonical 256-bit scalar (the reference's `G1 * Fr`)
__device__ __noinline__ G1XYZZ g1_scalar_mul(const G1Affine& base, const U256& s) {
    G1XYZZ acc = G1XYZZ::identity();
    U256 k = s;
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        acc = xyzz_double(acc);
        const uint32_t top = k.w[7] >> 31;
#pragma unroll
        for (int i = 7; i > 0; --i) k.w[i] = (k.w[i] << 1) | (k.w[i - 1] >> 31);
        k.w[0] <<= 1;
        if (top) xyzz_add_affine(acc, base);
    }
    return acc;
}

// MockEccChip::scalar_mul / scalar_mul_constant  (mock/arith/ecc.rs:88-104)
__global__ void __launch_bounds__(BLOCK) k_g1_batch_scalar_mul(const uint8_t* __restrict__ bases,
                                                               const uint8_t* __restrict__ scalars, size_t n,
                                                               uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        uint32_t bad = 0;
        G1Affine p = affine_load_canonical(bases + 64 * i, bad);
        U256 s = u256_load(scalars + 32 * i);
        bad |= !u256_is_canonical_fr(s);
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        jac_store_canonical(out + 96 * i, jac_from_xyzz(g1_scalar_mul(p, s)));
    }
}

FP_INLINE G1Affine affine_from_xyzz(const G1XYZZ& p) {
    G1Affine r;
    if (p.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        return r;
    }
    // 1/ZZ = ZZZ^2 / ZZ^4 ... one inversion of ZZZ gives both: 1/ZZZ = i; 1/ZZ = i^2 * ZZ^2 ... use
    // i = 1/(ZZ*ZZZ): 1/ZZ = i*ZZZ, 1/ZZZ = i*ZZ.
    Fq i = fp_inv<FqParams>(FQ_MUL(p.zz, p.zzz));
    r.x = FQ_MUL(p.x, FQ_MUL(i, p.zzz));
    r.y = FQ_MUL(p.y, FQ_MUL(i, p.zz));
    return r;
}

// MockEccChip::to_value = to_affine  (mock/arith/ecc.rs:64-66)
// One inversion per point (safegcd, ~13 k instructions) is 3/4 of a point's work, so a lane shares one inversion among
// TA_K of its points with Montgomery's trick (prefix products of the z's kept in registers, points re-read on the way
// back: the kernel is far from bandwidth-bound): 2.8 ms -> see profiles/r02_final_batch_roofline.txt for 2^22 points.
constexpr int TA_K = 8;
// MONT = false: canonical affine out (the C ABI's to_affine); true: Montgomery affine out, the form of a base table —
// h2agg_g1_msm_jac normalises the caller's projective points on the device instead of asking for batch_normalize on the host
template <bool MONT>
FP_INLINE void jac_batch_normalise(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out, uint32_t* flags) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i0 = (size_t)blockIdx.x * BLOCK + threadIdx.x; i0 < n; i0 += stride * TA_K) {
        // points i0, i0 + stride, ... (consecutive lanes touch consecutive points in every step)
        Fq pre[TA_K];
        Fq run = Fq::one();
        uint32_t bad = 0;
#pragma unroll
        for (int j = 0; j < TA_K; ++j) {
            const size_t i = i0 + (size_t)j * stride;
            Fq z = Fq::one();
            if (i < n) {
                bad |= jac_noncanonical(in + 96 * i);
                const Fq zc = fp_to_mont<FqParams>(fp_load<FqParams>(in + 96 * i + 64));
                if (!fp_is_zero_mod<2, FqParams>(zc)) z = zc;               // identity: z = 0 stays out of the product
            }
            pre[j] = run;                                                   // product of the z's before point j
            run = FQ_MUL(run, z);
        }
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        Fq inv = fp_inv<FqParams>(run);                                     // 1 / (z_0 ... z_{K-1})
#pragma unroll
        for (int j = TA_K - 1; j >= 0; --j) {
            const size_t i = i0 + (size_t)j * stride;
            if (i >= n) continue;
            const Fq zc = fp_to_mont<FqParams>(fp_load<FqParams>(in + 96 * i + 64));
            const bool ident = fp_is_zero_mod<2, FqParams>(zc);
            Fq ox = Fq::zero(), oy = Fq::zero();
            if (!ident) {
                const Fq zi = FQ_MUL(inv, pre[j]);                           // 1 / z_j
                inv = FQ_MUL(inv, zc);                                      // drop z_j from the running inverse
                const Fq zi2 = FQ_SQR(zi);
                const Fq x = fp_to_mont<FqParams>(fp_load<FqParams>(in + 96 * i));
                const Fq y = fp_to_mont<FqParams>(fp_load<FqParams>(in + 96 * i + 32));
                ox = FQ_MUL(x, zi2);                                        // x / z^2
                oy = FQ_MUL(y, FQ_MUL(zi2, zi));                            // y / z^3
                if (!MONT) {
                    ox = fp_from_mont<FqParams>(ox);
                    oy = fp_from_mont<FqParams>(oy);
                }
            }
            fp_store<FqParams>(out + 64 * i, ox);
            fp_store<FqParams>(out + 64 * i + 32, oy);
        }
    }
}
__global__ void __launch_bounds__(BLOCK) k_g1_batch_to_affine(const uint8_t* __restrict__ in, size_t n,
                                                              uint8_t* __restrict__ out, uint32_t* flags) {
    jac_batch_normalise<false>(in, n, out, flags);
}
__global__ void __launch_bounds__(BLOCK) k_jac_to_mont_affine(const uint8_t* __restrict__ in, size_t n,
                                                              uint8_t* __restrict__ out, uint32_t* flags) {
    jac_batch_normalise<true>(in, n, out, flags);
}

// ------------------------------------------------------------------ proof wire format of a G1 point
// What the transcript reader hands to the path (systems/halo2/transcript.rs:56-79: read_exact(32 bytes) ->
// C::from_bytes -> "invalid point encoding in proof").  Encoding (halo2curves 0.2.1 GroupEncoding for bn256, recalled
// from upstream — SURVEY.md appendix C; the crate is not vendored in the reference): 32 bytes little-endian x with the
// parity of y (y.to_bytes()[0] & 1) in bit 7 of byte 31; the identity is 32 zero bytes.
// Decoding: x must be canonical (< p); x = 0 without the sign bit is the identity; otherwise y = sqrt(x^3 + 3) must exist
// (p = 3 mod 4: y = rhs^((p+1)/4), valid iff y^2 = rhs) and the root with the encoded parity is taken.
FP_INLINE Fq fq_sqrt_candidate(const Fq& a) {   // a^((p+1)/4), Montgomery in / out
    constexpr uint32_t E[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                               0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};   // (p + 1) / 4, 252 bits
    Fq acc = Fq::one();
#pragma unroll 1
    for (int bit = 251; bit >= 0; --bit) {
        acc = FQ_SQR(acc);
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) w = (bit / 32 == i) ? E[i] : w;
        if ((w >> (bit % 32)) & 1u) acc = FQ_MUL(acc, a);
    }
    return acc;
}
// one 32-byte encoding -> canonical affine coordinates (zeros for the identity and for an invalid encoding); false = invalid
FP_INLINE bool g1_decompress_one(U256 w, Fq& ox, Fq& oy) {
    const uint32_t ysign = w.w[7] >> 31;
    w.w[7] &= 0x7fffffffu;
    Fq x = fp_unpack<FqParams>(w.w);
    bool good = fp_is_canonical<FqParams>(x);
    ox = Fq::zero();
    oy = Fq::zero();
    if (good && !(x.is_zero_int() && !ysign)) {
        const Fq xm = fp_to_mont<FqParams>(x);
        Fq three;
#pragma unroll
        for (int k = 0; k < NL; ++k) three.l[k] = 0;
        three.l[0] = 3;
        const Fq rhs = FQ_ADD(FQ_MUL(FQ_SQR(xm), xm), fp_to_mont<FqParams>(three));   // x^3 + 3, [4]
        const Fq y = fq_sqrt_candidate(rhs);
        good = fp_is_zero_mod<8, FqParams>(FQ_SUB(4, FQ_SQR(y), rhs));               // y^2 == rhs
        Fq yc = fp_from_mont<FqParams>(y);                                             // canonical
        if ((yc.l[0] & 1u) != ysign) {                                                 // take the other root: p - y
            int32_t d[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) d[k] = (int32_t)FqParams::MOD[k] - (int32_t)yc.l[k];
            yc = fp_normalize<FqParams>(d);                                            // y != 0 here (3 is a non-residue)
        }
        if (good) {
            ox = x;
            oy = yc;
        }
    }
    return good;
}
__global__ void __launch_bounds__(BLOCK) k_g1_batch_decompress(const uint8_t* __restrict__ in, size_t n,
                                                               uint8_t* __restrict__ out_aff, uint8_t* __restrict__ ok,
                                                               uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Fq ox, oy;
        const bool good = g1_decompress_one(u256_load(in + 32 * i), ox, oy);
        if (!good) atomicOr(flags, FLAG_BAD_POINT);
        if (ok) ok[i] = good ? 1 : 0;
        fp_store<FqParams>(out_aff + 64 * i, ox);
        fp_store<FqParams>(out_aff + 64 * i + 32, oy);
    }
}
// canonical affine (64 B, identity = zeros) -> 32-byte encoding
__global__ void __launch_bounds__(BLOCK) k_g1_batch_compress(const uint8_t* __restrict__ aff, size_t n,
                                                             uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Fq x = fp_load<FqParams>(aff + 64 * i), y = fp_load<FqParams>(aff + 64 * i + 32);
        if (!fp_is_canonical<FqParams>(x) | !fp_is_canonical<FqParams>(y)) atomicOr(flags, FLAG_NONCANONICAL);
        uint32_t w[8];
        fp_pack<FqParams>(w, x);
        if (!(x.is_zero_int() && y.is_zero_int())) w[7] |= (y.l[0] & 1u) << 31;
        uint4* o = reinterpret_cast<uint4*>(out + 32 * i);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// sum of n Jacobian points, one workgroup
__global__ void __launch_bounds__(BLOCK) k_g1_sum(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out,
                                                  uint32_t* flags) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    G1XYZZ acc = G1XYZZ::identity();
    for (size_t i = threadIdx.x; i < n; i += BLOCK) {
        if (jac_noncanonical(in + 96 * i)) atomicOr(flags, FLAG_NONCANONICAL);
        acc = xyzz_add(acc, xyzz_from_jac(jac_load_canonical(in + 96 * i)));
    }
    G1XYZZ tot = block_sum_xyzz(acc, lds);
    if (threadIdx.x == 0) jac_store_canonical(out, jac_from_xyzz(tot));
}

// y^2 = x^3 + 3 z^6 (or z = 0): partial accumulators arrive over a transport and are checked like every other input
FP_INLINE bool jac_on_curve(const G1Jac& p) {
    if (fp_is_zero_mod<2, FqParams>(p.z)) return true;
    Fq three;
#pragma unroll
    for (int k = 0; k < NL; ++k) three.l[k] = 0;
    three.l[0] = 3;
    const Fq z2 = FQ_SQR(p.z);
    const Fq z6 = FQ_MUL(FQ_SQR(z2), z2);
    const Fq rhs = FQ_ADD(FQ_MUL(FQ_SQR(p.x), p.x), FQ_MUL(z6, fp_to_mont<FqParams>(three)));   // [4]
    return fp_is_zero_mod<8, FqParams>(FQ_SUB(4, FQ_SQR(p.y), rhs));
}

// out_aff[k] = to_affine( sum_r in[(r * npts + k)] ), r < world: the local fold after the all-gather of the ranks' partial
// accumulators (one workgroup per output point; arithmetic = MockEccChip::add + to_value, mock/arith/ecc.rs:30-37,64-66)
__global__ void __launch_bounds__(BLOCK) k_g1_sum_strided_affine(const uint8_t* __restrict__ in, size_t world, size_t npts,
                                                                 uint8_t* __restrict__ out_aff, uint32_t* flags) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    const size_t k = blockIdx.x;
    G1XYZZ acc = G1XYZZ::identity();
    for (size_t r = threadIdx.x; r < world; r += BLOCK) {
        const uint8_t* p = in + 96 * (r * npts + k);
        if (jac_noncanonical(p)) atomicOr(flags, FLAG_NONCANONICAL);
        const G1Jac j = jac_load_canonical(p);
        if (!jac_on_curve(j)) atomicOr(flags, FLAG_BAD_POINT);
        acc = xyzz_add(acc, xyzz_from_jac(j));
    }
    G1XYZZ tot = block_sum_xyzz(acc, lds);
    if (threadIdx.x == 0) {
        const G1Affine a = affine_from_xyzz(tot);
        fp_store<FqParams>(out_aff + 64 * k, fp_from_mont<FqParams>(a.x));
        fp_store<FqParams>(out_aff + 64 * k + 32, fp_from_mont<FqParams>(a.y));
    }
}

// ------------------------------------------------------------------ base tables (device resident, Montgomery)
__global__ void __launch_bounds__(BLOCK) k_bases_to_mont(const uint8_t* __restrict__ in, size_t n,
                                                         uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        uint32_t bad = 0;
        G1Affine p = affine_load_canonical(in + 64 * i, bad);
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        affine_store(out + 64 * i, p);
    }
}
__global__ void __launch_bounds__(BLOCK) k_bases_from_mont(const uint8_t* __restrict__ in, size_t n,
                                                           uint8_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        G1Affine p = affine_load(in + 64 * i);
        fp_store<FqParams>(out + 64 * i, fp_from_mont<FqParams>(p.x));
        fp_store<FqParams>(out + 64 * i + 32, fp_from_mont<FqParams>(p.y));
    }
}
// bases[i] = k_i * G   (scalar_mul_constant with the generator, then to_affine; Montgomery output)
__global__ void __launch_bounds__(BLOCK) k_bases_generate(const uint8_t* __restrict__ k, size_t n,
                                                          uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        U256 s = u256_load(k + 32 * i);
        if (!u256_is_canonical_fr(s)) atomicOr(flags, FLAG_NONCANONICAL);
        G1Affine g;
        g.x = Fq::one();
        g.y = FQ_DBL(Fq::one());
        affine_store(out + 64 * i, affine_from_xyzz(g1_scalar_mul(g, s)));
    }
}

}  // namespace h2agg
