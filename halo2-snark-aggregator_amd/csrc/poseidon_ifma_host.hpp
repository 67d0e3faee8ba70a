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
#pragma once
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#define H2AGG_HAVE_IFMA_BUILD 1
#include <immintrin.h>

#include "poseidon_sponge_host.hpp"

namespace h2agg {
namespace poseidon_host {
namespace ifma {

#define IFMA_FN __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw"), always_inline)) static inline
#define IFMA_BIG __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw"), noinline)) static

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
    std::vector<std::array<uint64_t, 40>> R, A, D;   // [k][5][8]: R_k lanes, A_k lanes, D_k replicated
    uint64_t finBeta[5][8], finCum[5][8];
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
        for (int k = 0; k < r_p; ++k) {
            std::array<uint64_t, 40> r{}, a{}, d{};
            for (int i = 0; i < 8; ++i) {
                put_lane((uint64_t(*)[8])r.data(), i, s.ps_r[k][i]);
                put_lane((uint64_t(*)[8])a.data(), i, s.ps_a[k][i]);
            }
            put_all((uint64_t(*)[8])d.data(), s.ps_d[k]);
            R.push_back(r);
            A.push_back(a);
            D.push_back(d);
        }
        put_all(finBeta, s.ps_fin[0]);
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
    // partial rounds, scaled: W = w (s0 = beta_k w), V = shat
#pragma GCC unroll 1
    for (int k = 0; k < C.r_p; ++k) {
        A10 d = zero10();
        mul_acc(d, load5((const uint64_t(*)[8])C.R[k].data()), V);
        hsum(d);
        add_shifted(d, load5((const uint64_t(*)[8])C.D[k].data()));
        const V5 e = redc(d, C);                   // D_k + sum_i R_{k,i} shat_i   (off the chain)
        const V5 z = pow5(W, C);
        W = vadd(z, e);
        A10 u = zero10();
        mul_acc(u, load5((const uint64_t(*)[8])C.A[k].data()), z);
        add_shifted(u, V);
        V = redc(u, C);                            // shat_i + A_{k,i} z
        // a reduction only promises "< exact + r": with shat folded into the product the slack would add up round after
        // round (65 r < 2^260 at the end, too close); a multiplication by one every 16 rounds resets it
        if ((k & 15) == 15) V = vmul(V, load5(C.oneV), C);
    }
    W = vmul(W, load5(C.finBeta), C);              // s0 = beta_63 w
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
__attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw"))) static bool sponge_run(
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
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw");
}

}  // namespace ifma
}  // namespace poseidon_host
}  // namespace h2agg
#endif
