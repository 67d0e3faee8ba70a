// The host sponge's permutation on AVX-512 IFMA (vpmadd52luq / vpmadd52huq): eight field elements per vector.
//
// Same permutation, same constants, same results as csrc/poseidon_sponge_host.hpp (PoseidonChip::permutation,
// halo2-snark-aggregator-api/src/hash/poseidon.rs:193-230, T = 9, R_F = 8, R_P = 63); selected at run time when the CPU has
// avx512ifma (the MI355X boxes' EPYC 9575F does), the 4 x 64-bit code stays as the portable path and as the differential
// partner (tests/test_host_sponge.py runs both).
//
// Representation: 5 limbs x 52 bits, Montgomery with R' = 2^260; a vector set V5 holds limb i of eight elements in l[i].
// A product is 25 lo + 25 hi multiply-adds into ten 64-bit accumulators per lane (sums of up to 90 terms < 2^52: no overflow),
// Montgomery reduction is word-serial on the accumulators (5 x (1 + 10) multiply-adds), carries are propagated with shifts.
// Values are kept < 2^260 with normalised limbs, lazily reduced (R'/r ~ 84: a product of inputs < A r, < B r is
// < (A B / 84 + 1) r), and made canonical only when a challenge leaves the sponge.
//
// State layout: s[1..8] in the eight lanes of V; s[0] REPLICATED in every lane of W — the butterfly that sums a dot product
// across lanes leaves the total in every lane, so the word the partial rounds keep multiplying into the other eight never
// needs a broadcast.  Partial rounds run in the scaled form of Spec::scale_partial_rounds (csrc/poseidon_host.hpp), like the
// device kernel: per round  z = w^5;  shat += A_k z;  w <- z + D_k + sum_i R_{k,i} shat_i  (the sum uses the shat of BEFORE the
// update and is off the w -> z -> w chain).
//
// Round 5: the chain of the partial rounds leaves the vectors.  A permutation's latency is its 63 partial rounds' w -> w^5 -> w
// chain, three dependent products per round, and a vector product here is ~90 cycles deep (25 + 25 multiply-adds into ten
// accumulators, a word-serial reduction, carries) however wide it is.  The chain is ONE element: it now runs on the scalar
// unit — 4 x 64-bit "no-carry" CIOS over mulx with two carry chains (BMI2 + ADX; the arithmetic of csrc/pairing.hpp's second
// build), values kept below 2 r without any conditional subtraction inside w^5 (r / 2^256 = 0.19: inputs < 2 r give
// < (4 x 0.19 + 1) r) — while everything that is eight wide (shat += A_k z; D_k + sum_i R_{k,i} shat_i) stays on IFMA beside
// it, off the chain.  The two Montgomery radices (2^256 there, 2^260 here) meet in the constants: R_k and D_k are stored
// divided by 16, so the sum comes out of the vector reduction in the scalar side's form; A_k and beta_63 multiplied by 16, so
// the scalar z goes into the vector products as it is.  Same integers at the end (tests/test_host_sponge.py: both kernels
// against each other and against the oracle).  17.5 -> ~8 us per permutation on the build container's Xeon.
#pragma once
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#define H2AGG_HAVE_IFMA_BUILD 1
#include <immintrin.h>

#include "poseidon_sponge_host.hpp"

namespace h2agg {
namespace poseidon_host {
namespace ifma {

#define IFMA_FN __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw,bmi2,adx"), always_inline)) static inline
#define IFMA_BIG __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw,bmi2,adx"), noinline)) static

struct V5 {
    __m512i l[5];
};
struct A10 {
    __m512i t[10];
};
constexpr uint64_t M52 = ((uint64_t)1 << 52) - 1;

// ---- scalar helpers (setup and I/O only) --------------------------------------------------------------------------
// canonical integer (4 x 64) -> five 52-bit limbs
static inline void words_to_limbs(const uint64_t w[4], uint64_t l[5]) {
    l[0] = w[0] & M52;
    l[1] = ((w[0] >> 52) | (w[1] << 12)) & M52;
    l[2] = ((w[1] >> 40) | (w[2] << 24)) & M52;
    l[3] = ((w[2] >> 28) | (w[3] << 36)) & M52;
    l[4] = w[3] >> 16;
}
static inline void limbs_to_words(const uint64_t l[5], uint64_t w[4]) {
    w[0] = l[0] | (l[1] << 52);
    w[1] = (l[1] >> 12) | (l[2] << 40);
    w[2] = (l[2] >> 24) | (l[3] << 28);
    w[3] = (l[3] >> 36) | (l[4] << 16);
}
// a value held in the 4 x 64 code's Montgomery form (x 2^256) -> limbs of its R' = 2^260 form (x 2^260 mod r)
static inline void mont256_to_limbs(const HFr& a, uint64_t l[5]) {
    const uint64_t sixteen[4] = {16, 0, 0, 0};
    // a.l is the INTEGER x * 2^256 mod r; times 16 mod r is x * 2^260 mod r
    const HFr d = from_words(a.l);
    uint64_t w[4];
    to_canonical(mul(d, from_words(sixteen)), w);
    words_to_limbs(w, l);
}

struct Consts {
    // lane-replicated modulus limbs and -r^-1 mod 2^52
    uint64_t p[5], ninv;
    // every vector constant: [5 limbs][8 lanes]
    typedef uint64_t VC[5][8];
    VC pcV, pcW, oneV;                       // pre-constants (start[0]); Montgomery one
    std::vector<std::array<uint64_t, 80>> startV, startW, endV, endW;   // [k]: start[k] k = 1..h, end[k]
    // dense matrices: col[j] = lanes i -> M[i + 1][j] (j = 0..8), row0V = lanes j -> M[0][j + 1], m00 replicated
    struct Dense {
        uint64_t col[9][5][8], row0[5][8], m00[5][8];
    } mds, pre;
    // scaled partial rounds
    // [k][5][8]: R_k / 16 lanes, 16 A_k lanes, D_k / 16 replicated — the factors of 16 = 2^260 / 2^256 that let the scalar chain
    // (Montgomery radix 2^256) and the vectors (2^260) exchange values without a conversion product (header)
    std::vector<std::array<uint64_t, 40>> R, A;   // [k][5][8]: 2^56 R_k lanes (see dot_words), 16 A_k lanes
    std::vector<HFr> Dk, Ck;                      // D_k; 2^64 <R_k, A_{k-1}> (C_0 = 0): scalar side's Montgomery form
    uint64_t finBeta[5][8], finCum[5][8];    // 16 beta_63 replicated; cum lanes
    HFr two252;                              // the integer 2^252 in plain words: (w 2^260)(2^252) / 2^256 = w 2^256
    int h = 0, r_p = 0;
    bool ok = false;

    static void put_lane(uint64_t (*dst)[8], int lane, const HFr& v) {
        uint64_t l[5];
        mont256_to_limbs(v, l);
        for (int i = 0; i < 5; ++i) dst[i][lane] = l[i];
    }
    static void put_all(uint64_t (*dst)[8], const HFr& v) {
        for (int lane = 0; lane < 8; ++lane) put_lane(dst, lane, v);
    }
    explicit Consts(const Spec& s) {
        if (!s.ok || s.t != 9) return;
        h = s.r_f / 2;
        r_p = s.r_p;
        {
            uint64_t l[5];
            words_to_limbs(R_MOD, l);
            for (int i = 0; i < 5; ++i) p[i] = l[i];
            // -r^-1 mod 2^52 by Newton iteration on the low limb (r odd)
            uint64_t inv = 1;
            for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
            ninv = (0 - inv) & M52;
        }
        auto vec9 = [&](const std::vector<HFr>& v, uint64_t (*V)[8], uint64_t (*W)[8]) {
            put_all(W, v[0]);
            for (int i = 1; i < 9; ++i) put_lane(V, i - 1, v[i]);
        };
        vec9(s.start[0], pcV, pcW);
        put_all(oneV, one());
        auto as2d = [](std::array<uint64_t, 80>& a, bool second) { return (uint64_t(*)[8])(a.data() + (second ? 40 : 0)); };
        (void)as2d;
        for (int k = 1; k <= h; ++k) {
            std::array<uint64_t, 80> v{}, w{};
            vec9(s.start[k], (uint64_t(*)[8])v.data(), (uint64_t(*)[8])w.data());
            startV.push_back(v);
            startW.push_back(w);
        }
        for (size_t k = 0; k < s.end.size(); ++k) {
            std::array<uint64_t, 80> v{}, w{};
            vec9(s.end[k], (uint64_t(*)[8])v.data(), (uint64_t(*)[8])w.data());
            endV.push_back(v);
            endW.push_back(w);
        }
        auto dense = [&](const Mat& m, Dense& d) {
            for (int j = 0; j < 9; ++j)
                for (int i = 1; i < 9; ++i) put_lane(d.col[j], i - 1, m[i][j]);
            for (int j = 1; j < 9; ++j) put_lane(d.row0, j - 1, m[0][j]);
            put_all(d.m00, m[0][0]);
        };
        dense(s.mds, mds);
        dense(s.pre_sparse, pre);
        const uint64_t w16[4] = {16, 0, 0, 0}, w56[4] = {(uint64_t)1 << 56, 0, 0, 0}, w64[4] = {0, 1, 0, 0};
        const HFr f16 = from_words(w16), f56 = from_words(w56), f64 = from_words(w64);
        for (int k = 0; k < r_p; ++k) {
            std::array<uint64_t, 40> r{}, a{};
            HFr ck = zero();
            for (int i = 0; i < 8; ++i) {
                put_lane((uint64_t(*)[8])r.data(), i, mul(s.ps_r[k][i], f56));
                put_lane((uint64_t(*)[8])a.data(), i, mul(s.ps_a[k][i], f16));
                if (k) ck = add(ck, mul(s.ps_r[k][i], s.ps_a[k - 1][i]));
            }
            R.push_back(r);
            A.push_back(a);
            Dk.push_back(s.ps_d[k]);
            Ck.push_back(mul(ck, f64));
        }
        put_all(finBeta, mul(s.ps_fin[0], f16));
        two252 = HFr{{0, 0, 0, (uint64_t)1 << 60}};
        for (int i = 0; i < 8; ++i) put_lane(finCum, i, s.ps_fin[i + 1]);
        ok = true;
    }
};

// ---- vector arithmetic ----------------------------------------------------------------------------------------------
IFMA_FN V5 load5(const uint64_t (*c)[8]) {
    V5 r;
    for (int i = 0; i < 5; ++i) r.l[i] = _mm512_loadu_si512((const void*)c[i]);
    return r;
}
IFMA_FN A10 zero10() {
    A10 a;
    for (int i = 0; i < 10; ++i) a.t[i] = _mm512_setzero_si512();
    return a;
}
// acc += a * b
IFMA_FN void mul_acc(A10& acc, const V5& a, const V5& b) {
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) {
#pragma GCC unroll 5
        for (int j = 0; j < 5; ++j) {
            acc.t[i + j] = _mm512_madd52lo_epu64(acc.t[i + j], a.l[i], b.l[j]);
            acc.t[i + j + 1] = _mm512_madd52hi_epu64(acc.t[i + j + 1], a.l[i], b.l[j]);
        }
    }
}
// acc += a * a (cross terms once, doubled)
IFMA_FN void sqr_acc(A10& acc, const V5& a) {
    A10 x = zero10();
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) {
#pragma GCC unroll 5
        for (int j = i + 1; j < 5; ++j) {
            x.t[i + j] = _mm512_madd52lo_epu64(x.t[i + j], a.l[i], a.l[j]);
            x.t[i + j + 1] = _mm512_madd52hi_epu64(x.t[i + j + 1], a.l[i], a.l[j]);
        }
    }
#pragma GCC unroll 10
    for (int k = 0; k < 10; ++k) acc.t[k] = _mm512_add_epi64(acc.t[k], _mm512_add_epi64(x.t[k], x.t[k]));
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) {
        acc.t[2 * i] = _mm512_madd52lo_epu64(acc.t[2 * i], a.l[i], a.l[i]);
        acc.t[2 * i + 1] = _mm512_madd52hi_epu64(acc.t[2 * i + 1], a.l[i], a.l[i]);
    }
}
// acc += v * 2^260  (v joins a Montgomery product as a plain addend: redc(a * b + v R') = a (x) b + v)
IFMA_FN void add_shifted(A10& acc, const V5& v) {
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) acc.t[5 + i] = _mm512_add_epi64(acc.t[5 + i], v.l[i]);
}
// Montgomery reduction by R' = 2^260: value(t) / 2^260 mod r, normalised limbs, < value(t) / 2^260 + r
IFMA_FN V5 redc(A10 t, const Consts& C) {
    const __m512i mask = _mm512_set1_epi64((long long)M52), ninv = _mm512_set1_epi64((long long)C.ninv), zero = _mm512_setzero_si512();
    __m512i p[5];
    for (int j = 0; j < 5; ++j) p[j] = _mm512_set1_epi64((long long)C.p[j]);
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) {
        const __m512i m = _mm512_madd52lo_epu64(zero, t.t[i], ninv);
#pragma GCC unroll 5
        for (int j = 0; j < 5; ++j) {
            t.t[i + j] = _mm512_madd52lo_epu64(t.t[i + j], m, p[j]);
            t.t[i + j + 1] = _mm512_madd52hi_epu64(t.t[i + j + 1], m, p[j]);
        }
        t.t[i + 1] = _mm512_add_epi64(t.t[i + 1], _mm512_srli_epi64(t.t[i], 52));
    }
    V5 r;
    __m512i c = t.t[5];
#pragma GCC unroll 4
    for (int k = 0; k < 4; ++k) {
        r.l[k] = _mm512_and_si512(c, mask);
        c = _mm512_add_epi64(t.t[6 + k], _mm512_srli_epi64(c, 52));
    }
    r.l[4] = c;
    return r;
}
IFMA_FN V5 vmul(const V5& a, const V5& b, const Consts& C) {
    A10 t = zero10();
    mul_acc(t, a, b);
    return redc(t, C);
}
IFMA_FN V5 vsqr(const V5& a, const Consts& C) {
    A10 t = zero10();
    sqr_acc(t, a);
    return redc(t, C);
}
// a + b, limbs normalised (value < 2^260 by the callers' bounds)
IFMA_FN V5 vadd(const V5& a, const V5& b) {
    const __m512i mask = _mm512_set1_epi64((long long)M52);
    V5 r;
    __m512i c = _mm512_add_epi64(a.l[0], b.l[0]);
#pragma GCC unroll 4
    for (int k = 0; k < 4; ++k) {
        r.l[k] = _mm512_and_si512(c, mask);
        c = _mm512_add_epi64(_mm512_add_epi64(a.l[k + 1], b.l[k + 1]), _mm512_srli_epi64(c, 52));
    }
    r.l[4] = c;
    return r;
}
// x^5 + c
IFMA_FN V5 pow5_plus(const V5& x, const V5& c, const Consts& C) {
    const V5 x2 = vsqr(x, C);
    const V5 x4 = vsqr(x2, C);
    A10 t = zero10();
    mul_acc(t, x4, x);
    add_shifted(t, c);
    return redc(t, C);
}
IFMA_FN V5 pow5(const V5& x, const Consts& C) {
    const V5 x2 = vsqr(x, C);
    const V5 x4 = vsqr(x2, C);
    return vmul(x4, x, C);
}
// every lane <- the sum over the eight lanes (butterfly)
IFMA_FN void hsum(A10& a) {
#pragma GCC unroll 10
    for (int k = 0; k < 10; ++k) {
        __m512i t = a.t[k];
        t = _mm512_add_epi64(t, _mm512_shuffle_i64x2(t, t, 0x4E));
        t = _mm512_add_epi64(t, _mm512_shuffle_i64x2(t, t, 0xB1));
        t = _mm512_add_epi64(t, _mm512_shuffle_epi32(t, (_MM_PERM_ENUM)0x4E));
        a.t[k] = t;
    }
}
IFMA_FN V5 bcast_lane(const V5& v, int lane) {
    const __m512i idx = _mm512_set1_epi64(lane);
    V5 r;
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) r.l[i] = _mm512_permutexvar_epi64(idx, v.l[i]);
    return r;
}
// (W, V) <- M (W, V) for a dense 9 x 9 matrix
IFMA_BIG void dense(const Consts::Dense& M, V5& W, V5& V, const Consts& C) {
    V5 Wn;
    {
        A10 aw = zero10();
        mul_acc(aw, load5(M.row0), V);
        hsum(aw);
        mul_acc(aw, load5(M.m00), W);
        Wn = redc(aw, C);
    }
    A10 av = zero10();
    mul_acc(av, load5(M.col[0]), W);
#pragma GCC unroll 1
    for (int j = 1; j < 9; ++j) mul_acc(av, load5(M.col[j]), bcast_lane(V, j - 1));
    V = redc(av, C);
    W = Wn;
}

// ---- the scalar side of the partial rounds: 4 x 64-bit Montgomery (radix 2^256), values < 4 r, no final subtraction -----
// a * b / 2^256 mod r for a, b < 4 r < 2^256: result < a b / 2^256 + r  (the running sum stays below b + r < 2^256)
IFMA_FN HFr smul_lazy(const HFr& a, const HFr& b) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, A, Cw, hi, lo, m;
    unsigned char c1, c2;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        lo = _mulx_u64(a.l[i], b.l[0], &hi); c1 = _addcarry_u64(0, lo, t0, &t0); A = hi;
        lo = _mulx_u64(a.l[i], b.l[1], &hi); c2 = _addcarry_u64(0, lo, A, &lo); A = hi; c1 = _addcarry_u64(c1, lo, t1, &t1);
        lo = _mulx_u64(a.l[i], b.l[2], &hi); c2 = _addcarry_u64(c2, lo, A, &lo); A = hi; c1 = _addcarry_u64(c1, lo, t2, &t2);
        lo = _mulx_u64(a.l[i], b.l[3], &hi); c2 = _addcarry_u64(c2, lo, A, &lo); A = hi; c1 = _addcarry_u64(c1, lo, t3, &t3);
        _addcarry_u64(c2, A, 0, &A);
        _addcarry_u64(c1, A, 0, &A);
        m = t0 * R_INV;
        lo = _mulx_u64(m, R_MOD[0], &hi); c2 = _addcarry_u64(0, lo, t0, &lo); Cw = hi;
        lo = _mulx_u64(m, R_MOD[1], &hi); c2 = _addcarry_u64(c2, Cw, lo, &lo); Cw = hi; c1 = _addcarry_u64(0, lo, t1, &t0);
        lo = _mulx_u64(m, R_MOD[2], &hi); c2 = _addcarry_u64(c2, Cw, lo, &lo); Cw = hi; c1 = _addcarry_u64(c1, lo, t2, &t1);
        lo = _mulx_u64(m, R_MOD[3], &hi); c2 = _addcarry_u64(c2, Cw, lo, &lo); Cw = hi; c1 = _addcarry_u64(c1, lo, t3, &t2);
        _addcarry_u64(c2, Cw, 0, &Cw);
        _addcarry_u64(c1, Cw, A, &t3);
    }
    return HFr{{t0, t1, t2, t3}};
}
// a + b (no overflow by the callers' bounds), then - 2 r if that does not go negative: a + b < 4 r -> result < 2 r
IFMA_FN HFr sadd_csub2r(const HFr& a, const HFr& b) {
    static const uint64_t R2[4] = {0x87c3eb27e0000002ull, 0x5067d090f372e122ull, 0x70a08b6d0302b0baull, 0x60c89ce5c2634053ull};   // 2 r
    unsigned long long s0, s1, s2, s3, d0, d1, d2, d3;
    unsigned char c = _addcarry_u64(0, a.l[0], b.l[0], &s0);
    c = _addcarry_u64(c, a.l[1], b.l[1], &s1);
    c = _addcarry_u64(c, a.l[2], b.l[2], &s2);
    (void)_addcarry_u64(c, a.l[3], b.l[3], &s3);
    unsigned char bw = _subborrow_u64(0, s0, R2[0], &d0);
    bw = _subborrow_u64(bw, s1, R2[1], &d1);
    bw = _subborrow_u64(bw, s2, R2[2], &d2);
    bw = _subborrow_u64(bw, s3, R2[3], &d3);
    const uint64_t keep = (uint64_t)0 - (uint64_t)bw;   // borrowed: the sum was below 2 r, keep it
    return HFr{{(s0 & keep) | (d0 & ~keep), (s1 & keep) | (d1 & ~keep), (s2 & keep) | (d2 & ~keep), (s3 & keep) | (d3 & ~keep)}};
}
// The dot product <R_k, shat> of a partial round leaves the vectors UNREDUCED: ten 64-bit accumulators per lane, summed over
// the lanes (hsum), lane 0 read out and packed into nine 64-bit words here; the scalar side adds its own share to it and
// reduces once, by 2^320 (five word steps: the vectors' lazily reduced inputs make the sum up to ~2^515, and 2^-320 brings any
// of that below 1.01 r).  The powers of two are in the constants: R_k x 2^56 in the vectors (2^56 R'^2 = R 2^320), C_k x 2^64.
IFMA_FN void acc10_lane0_words(const A10& a, uint64_t T[9]) {
    unsigned __int128 carry = 0;
    uint64_t t[10];
    for (int i = 0; i < 10; ++i) t[i] = (uint64_t)_mm_cvtsi128_si64(_mm512_castsi512_si128(a.t[i]));
    // word j collects the bits [64 j, 64 j + 64) of sum_i t[i] 2^(52 i)
#pragma GCC unroll 9
    for (int j = 0; j < 9; ++j) {
        unsigned __int128 v = carry;
#pragma GCC unroll 10
        for (int i = 0; i < 10; ++i) {
            const int lo = 52 * i - 64 * j;          // position of t[i]'s bit 0 relative to word j
            if (lo >= 64 || lo <= -64) continue;
            if (lo >= 0) v += (uint64_t)(t[i] << lo);   // (its bits above the word are the next word's `>>` share)
            else v += (unsigned __int128)(t[i] >> (-lo));
        }
        T[j] = (uint64_t)v;
        carry = v >> 64;
    }
}
// T += a * b  (4 x 4 words into nine)
IFMA_FN void wide_mul_add(uint64_t T[9], const HFr& a, const HFr& b) {
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        unsigned long long hi, lo, c = 0;
        unsigned char cy;
#pragma GCC unroll 4
        for (int j = 0; j < 4; ++j) {
            lo = _mulx_u64(a.l[i], b.l[j], &hi);
            cy = _addcarry_u64(0, lo, c, &lo);
            hi += cy;                                   // (hi <= 2^64 - 2: no overflow)
            unsigned long long tt;
            cy = _addcarry_u64(0, T[i + j], lo, &tt);
            T[i + j] = tt;
            c = hi + cy;
        }
        unsigned long long tt;
        cy = _addcarry_u64(0, T[i + 4], c, &tt);
        T[i + 4] = tt;
        for (int k = i + 5; cy && k < 9; ++k) {
            cy = _addcarry_u64(cy, T[k], 0, &tt);
            T[k] = tt;
        }
    }
}
// T / 2^320 mod r for T < 2^520: < T / 2^320 + r < 1.01 r
IFMA_FN HFr redc5(uint64_t T[9]) {
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) {
        const unsigned long long m = T[i] * R_INV;
        unsigned long long hi, lo, c = 0, tt;
        unsigned char cy;
#pragma GCC unroll 4
        for (int j = 0; j < 4; ++j) {
            lo = _mulx_u64(m, R_MOD[j], &hi);
            cy = _addcarry_u64(0, lo, c, &lo);
            hi += cy;
            cy = _addcarry_u64(0, T[i + j], lo, &tt);
            T[i + j] = tt;
            c = hi + cy;
        }
        cy = _addcarry_u64(0, T[i + 4], c, &tt);
        T[i + 4] = tt;
        for (int k = i + 5; k < 9; ++k) {
            cy = _addcarry_u64(cy, T[k], 0, &tt);
            T[k] = tt;
        }
    }
    return HFr{{T[5], T[6], T[7], T[8]}};
}
IFMA_FN HFr sadd(const HFr& a, const HFr& b) {   // no overflow by the callers' bounds
    unsigned long long s0, s1, s2, s3;
    unsigned char c = _addcarry_u64(0, a.l[0], b.l[0], &s0);
    c = _addcarry_u64(c, a.l[1], b.l[1], &s1);
    c = _addcarry_u64(c, a.l[2], b.l[2], &s2);
    (void)_addcarry_u64(c, a.l[3], b.l[3], &s3);
    return HFr{{s0, s1, s2, s3}};
}
// lane 0 of a vector value (< 2^256) as 4 x 64-bit words / a 4-word value (< 2^256) replicated into every lane's five limbs
IFMA_FN HFr lane0_words(const V5& v) {
    uint64_t l[5], w[4];
    for (int i = 0; i < 5; ++i) l[i] = (uint64_t)_mm_cvtsi128_si64(_mm512_castsi512_si128(v.l[i]));
    limbs_to_words(l, w);
    return HFr{{w[0], w[1], w[2], w[3]}};
}
IFMA_FN V5 bcast_words(const HFr& a) {
    uint64_t l[5];
    words_to_limbs(a.l, l);
    V5 r;
    for (int i = 0; i < 5; ++i) r.l[i] = _mm512_set1_epi64((long long)l[i]);
    return r;
}

struct State {
    V5 W, V;
};

// PoseidonChip::permutation; IN: the chunk's inputs in lanes 0..n_in-1 (R' form), zero elsewhere
IFMA_BIG void permute(const Consts& C, State& S, const V5& IN, int n_in) {
    V5 W = S.W, V = S.V;
    {   // absorb_with_pre_constants (hash/poseidon.rs:45-86)
        W = vadd(W, load5(C.pcW));
        V5 add = vadd(IN, load5(C.pcV));
        if (n_in < 8) {   // the padding one lands on s[n_in + 1] = lane n_in
            const __mmask8 k = (__mmask8)(1u << n_in);
            const V5 one = load5(C.oneV);
            V5 o;
            for (int i = 0; i < 5; ++i) o.l[i] = _mm512_maskz_mov_epi64(k, one.l[i]);
            add = vadd(add, o);
        }
        V = vadd(V, add);
    }
    for (int k = 1; k <= C.h; ++k) {
        W = pow5_plus(W, load5((const uint64_t(*)[8])C.startW[k - 1].data()), C);
        V = pow5_plus(V, load5((const uint64_t(*)[8])C.startV[k - 1].data()), C);
        dense(k < C.h ? C.mds : C.pre, W, V, C);
    }
    // partial rounds, scaled: w (s0 = beta_k w) on the scalar unit in radix-2^256 Montgomery form, < 2 r; V = shat
    // (w arrives from a vector reduction: < 2 r < 2^256; times 2^252 / 2^256 takes the 16 out of its radix)
    HFr w = smul_lazy(lane0_words(W), C.two252);
    // With S_k = shat before round k:  e_k = D_k + <R_k, S_k> = D_k + <R_k, S_{k-1}> + <R_k, A_{k-1}> z_{k-1}: the vectors' share
    // of e_k only needs S_{k-1}, a whole round earlier than z_{k-1} exists (pre, carried from the previous iteration); the
    // last-minute share is ONE scalar product with a constant.  So the vectors never sit on the w -> w^5 -> w chain.
    uint64_t pre[9];
    {
        A10 d = zero10();
        mul_acc(d, load5((const uint64_t(*)[8])C.R[0].data()), V);
        hsum(d);
        acc10_lane0_words(d, pre);
    }
    HFr zprev = HFr{{0, 0, 0, 0}};
#pragma GCC unroll 1
    for (int k = 0; k < C.r_p; ++k) {
        wide_mul_add(pre, C.Ck[k], zprev);         // + 2^64 <R_k, A_{k-1}> z_{k-1}   (C_0 = 0)
        const HFr e = sadd(redc5(pre), C.Dk[k]);   // D_k + <R_k, S_k>, < 2.01 r
        if (k + 1 < C.r_p) {                       // the vectors' share of e_{k+1}: from S_k, before z_k is known
            A10 d = zero10();
            mul_acc(d, load5((const uint64_t(*)[8])C.R[k + 1].data()), V);
            hsum(d);
            acc10_lane0_words(d, pre);
        }
        const HFr w2 = smul_lazy(w, w);            // < 1.76 r
        const HFr w4 = smul_lazy(w2, w2);          // < 1.6 r
        const HFr z = smul_lazy(w4, w);            // w^5 < 1.6 r
        w = sadd_csub2r(z, e);                     // < 3.7 r before, < 2 r after
        zprev = z;
        A10 u = zero10();
        mul_acc(u, load5((const uint64_t(*)[8])C.A[k].data()), bcast_words(z));
        add_shifted(u, V);
        V = redc(u, C);                            // S_{k+1} = S_k + A_k z_k
        // a reduction only promises "< exact + r": with shat folded into the product the slack would add up round after
        // round (65 r < 2^260 at the end, too close); a multiplication by one every 16 rounds resets it
        if ((k & 15) == 15) V = vmul(V, load5(C.oneV), C);
    }
    W = vmul(bcast_words(w), load5(C.finBeta), C); // s0 = beta_63 w, back in the vectors' form
    V = vadd(V, load5(C.finCum));                  // s_i = shat_i + cum_i
    for (size_t k = 0; k < C.endV.size(); ++k) {
        W = pow5_plus(W, load5((const uint64_t(*)[8])C.endW[k].data()), C);
        V = pow5_plus(V, load5((const uint64_t(*)[8])C.endV[k].data()), C);
        dense(C.mds, W, V, C);
    }
    W = pow5(W, C);
    V = pow5(V, C);
    dense(C.mds, W, V, C);
    S.W = W;
    S.V = V;
}

// the sponge over one element stream (see sponge_run in poseidon_sponge_host.hpp)
__attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw,bmi2,adx"))) static bool sponge_run(
    const Consts& C, const uint8_t* elems, const uint32_t* upto, uint32_t nsq, uint8_t* out) {
    State S;
    {
        alignas(64) uint64_t w0[5][8] = {}, z8[5][8] = {};
        const uint64_t two64[4] = {0, 1, 0, 0};   // poseidon::State::default(): (2^64, 0, ..., 0)
        Consts::put_all(w0, from_words(two64));
        S.W = load5(w0);
        S.V = load5(z8);
    }
    // canonical -> R' form: redc(x * (R'^2 mod r)) = x R'
    alignas(64) uint64_t rr[5][8];
    {
        // R'^2 mod r = 2^520 mod r: Montgomery-256 form of 2^264 is 2^264 * 2^256 = 2^520
        HFr a = one();                              // the integer 2^256 mod r
        for (int i = 0; i < 8; ++i) a = add(a, a);   // 2^264 mod r
        const HFr b = from_words(a.l);              // (2^264 mod r) * 2^256 mod r = 2^520 mod r, as an integer
        uint64_t l[5];
        words_to_limbs(b.l, l);
        for (int i = 0; i < 5; ++i)
            for (int lane = 0; lane < 8; ++lane) rr[i][lane] = l[i];
    }
    const V5 RR = load5(rr);
    bool canonical = true;
    uint32_t pos = 0;
    for (uint32_t q = 0; q < nsq; ++q) {
        const uint32_t end = upto[q];
        uint32_t padding_offset = 0;
        bool any = false;
        while (pos < end) {
            const int nin = (int)((end - pos) < 8u ? (end - pos) : 8u);
            alignas(64) uint64_t in[5][8] = {};
            for (int i = 0; i < nin; ++i) {
                uint64_t w[4], l[5];
                memcpy(w, elems + 32 * (size_t)(pos + i), 32);
                if (geq_mod(w)) canonical = false;
                words_to_limbs(w, l);
                for (int k = 0; k < 5; ++k) in[k][i] = l[k];
            }
            const V5 IN = vmul(load5(in), RR, C);
            permute(C, S, IN, nin);
            padding_offset = (uint32_t)(8 - nin);
            pos += (uint32_t)nin;
            any = true;
        }
        if (!any || padding_offset == 0) {
            alignas(64) uint64_t z8[5][8] = {};
            permute(C, S, load5(z8), 0);
        }
        // challenge = s[1] = lane 0 of V, out of Montgomery form, canonical
        alignas(64) uint64_t onei[5][8] = {};
        for (int lane = 0; lane < 8; ++lane) onei[0][lane] = 1;
        const V5 plain = vmul(S.V, load5(onei), C);
        alignas(64) uint64_t lim[5][8];
        for (int i = 0; i < 5; ++i) _mm512_store_si512((void*)lim[i], plain.l[i]);
        uint64_t l[5], w[4];
        for (int i = 0; i < 5; ++i) l[i] = lim[i][0];
        limbs_to_words(l, w);
        while (geq_mod(w)) sub_mod(w);
        memcpy(out + 32 * (size_t)q, w, 32);
    }
    return canonical;
}

static inline bool cpu_has_ifma() {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("bmi2") &&
           __builtin_cpu_supports("adx");
}

}  // namespace ifma
}  // namespace poseidon_host
}  // namespace h2agg
#endif
