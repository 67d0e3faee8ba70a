// Poseidon sponge on HOST threads: the transcript backend for small batches.
//
// A proof's transcript is ONE dependent chain of ~136 permutations (T = 9, R_F = 8, R_P = 63): nothing inside it is
// data-parallel, so on the device it runs at a lone wave's latency (~160 us per permutation, csrc/poseidon_kernels.hpp),
// whatever the batch size up to thousands of proofs.  A host core does the same permutation in a few microseconds; with one
// thread per proof the host wins until the batch outgrows the cores (crossover measured in profiles/r03_sweeps.txt), and the
// device sponge keeps the large batches.  Both backends read the SAME generated constants (poseidon_host::Spec) and the same
// element streams (k_transcript_elements: point decompression and PoseidonEncode stay on the device), so the challenges are
// bit-identical (tests/test_gpu_verifier.py runs both).
//
// Stands behind
//   PoseidonChip::{update, squeeze, permutation}   halo2-snark-aggregator-api/src/hash/poseidon.rs:144-231
//   PoseidonTranscriptRead::squeeze_challenge_scalar .../systems/halo2/transcript.rs:56-119
//   the per-proof and the aggregation transcripts  halo2-snark-aggregator-circuit/src/verify_circuit.rs:121-163
//
// Arithmetic: 4 x 64-bit Montgomery over Fr (R = 2^256).  The linear layers are dot products of 9 terms: their 512-bit
// products are summed UNREDUCED (9 r^2 < 2^512) and reduced once — 9 x 16 + 20 word multiplications instead of 9 x 36.
#pragma once
#include <stdint.h>
#include <string.h>

#include "poseidon_host.hpp"

namespace h2agg {
namespace poseidon_host {

#define PSD_INLINE static inline __attribute__((always_inline))
// (hi, lo) = a * b + c + d   (never overflows 128 bits)
PSD_INLINE uint64_t mac2(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t& hi) {
    const u128 t = (u128)a * b + c + d;
    hi = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
// acc (512 bit) += a * b.  The caller keeps the running total below 2^512 (at most nine products of values < r).
// Straight-line: the product goes into eight clean words first, then ONE eight-word carry chain adds it in.
PSD_INLINE void wide_mac(uint64_t acc[8], const HFr& a, const HFr& b) {
    uint64_t t[8], c;
    t[0] = mac2(a.l[0], b.l[0], 0, 0, c);
    t[1] = mac2(a.l[0], b.l[1], c, 0, c);
    t[2] = mac2(a.l[0], b.l[2], c, 0, c);
    t[3] = mac2(a.l[0], b.l[3], c, 0, c);
    t[4] = c;
#pragma GCC unroll 3
    for (int i = 1; i < 4; ++i) {
        t[i] = mac2(a.l[i], b.l[0], t[i], 0, c);
        t[i + 1] = mac2(a.l[i], b.l[1], t[i + 1], c, c);
        t[i + 2] = mac2(a.l[i], b.l[2], t[i + 2], c, c);
        t[i + 3] = mac2(a.l[i], b.l[3], t[i + 3], c, c);
        t[i + 4] = c;
    }
    unsigned char cy = 0;
#pragma GCC unroll 8
    for (int i = 0; i < 8; ++i) {
        const u128 v = (u128)acc[i] + t[i] + cy;
        acc[i] = (uint64_t)v;
        cy = (unsigned char)(v >> 64);
    }
}
// acc += a * 2^256  (adds a canonical-Montgomery value `a` to the dot product: a = REDC(a * R))
PSD_INLINE void wide_add_shifted(uint64_t acc[8], const HFr& a) {
    u128 c = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        c += (u128)acc[4 + i] + a.l[i];
        acc[4 + i] = (uint64_t)c;
        c >>= 64;
    }
}
// r = (top : v) - (take ? r : 0), branch-free; returns the new top
PSD_INLINE uint64_t cond_sub_mod(uint64_t v[4], uint64_t top) {
    uint64_t d[4];
    unsigned char b = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const u128 x = (u128)v[i] - R_MOD[i] - b;
        d[i] = (uint64_t)x;
        b = (unsigned char)((x >> 64) & 1);
    }
    const bool take = top || !b;   // (top : v) >= r
    const uint64_t m = (uint64_t)0 - (uint64_t)take;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) v[i] = (d[i] & m) | (v[i] & ~m);
    return take ? top - b : top;
}
// Montgomery reduction of a 512-bit value T < 2^512 - 2^510: T / 2^256 mod r, fully reduced
PSD_INLINE HFr wide_redc(const uint64_t t_in[8]) {
    uint64_t t[8], top = 0;
#pragma GCC unroll 8
    for (int i = 0; i < 8; ++i) t[i] = t_in[i];
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * R_INV;
        uint64_t c;
        (void)mac2(m, R_MOD[0], t[i], 0, c);
        t[i + 1] = mac2(m, R_MOD[1], t[i + 1], c, c);
        t[i + 2] = mac2(m, R_MOD[2], t[i + 2], c, c);
        t[i + 3] = mac2(m, R_MOD[3], t[i + 3], c, c);
        // carry into the words above (the last round's carry leaves the eight words: `top`)
        unsigned char cy = 0;
        u128 v = (u128)t[i + 4 < 8 ? i + 4 : 7] + c;   // i + 4 <= 7
        t[i + 4] = (uint64_t)v;
        cy = (unsigned char)(v >> 64);
#pragma GCC unroll 3
        for (int k = i + 5; k < 8; ++k) {
            v = (u128)t[k] + cy;
            t[k] = (uint64_t)v;
            cy = (unsigned char)(v >> 64);
        }
        top += cy;
    }
    HFr r = {{t[4], t[5], t[6], t[7]}};
    top = cond_sub_mod(r.l, top);   // T / R < 2.7 r: two rounds suffice
    top = cond_sub_mod(r.l, top);
    return r;
}
// a + b mod r for values < r, branch-free
PSD_INLINE HFr fadd(const HFr& a, const HFr& b) {
    HFr r;
    unsigned char cy = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const u128 v = (u128)a.l[i] + b.l[i] + cy;
        r.l[i] = (uint64_t)v;
        cy = (unsigned char)(v >> 64);
    }
    (void)cond_sub_mod(r.l, cy);
    return r;
}
PSD_INLINE HFr fmul(const HFr& a, const HFr& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    wide_mac(t, a, b);
    return wide_redc(t);
}
PSD_INLINE HFr pow5(const HFr& x) {
    const HFr x2 = fmul(x, x);
    return fmul(fmul(x2, x2), x);
}

// the constants of one (T, R_F, R_P) instance in flat arrays (T <= 9)
struct FastSpec {
    static constexpr int MAXT = 9;
    int t = 0, h = 0, r_p = 0;
    std::vector<std::array<HFr, MAXT>> start, end, srow;   // [h + 1], [h - 1], [r_p]: srow[k][0] = row_0, [1..] = row_i
    std::vector<std::array<HFr, MAXT>> scol;               // [r_p][t - 1]
    std::vector<HFr> partial;
    HFr mds[MAXT][MAXT], pre[MAXT][MAXT];
    HFr one_m;                                             // Montgomery 1
    explicit FastSpec(const Spec& s) : t(s.t), h(s.r_f / 2), r_p(s.r_p) {
        auto row = [&](const std::vector<HFr>& v) {
            std::array<HFr, MAXT> a{};
            for (size_t i = 0; i < v.size(); ++i) a[i] = v[i];
            return a;
        };
        for (const auto& v : s.start) start.push_back(row(v));
        for (const auto& v : s.end) end.push_back(row(v));
        for (const auto& v : s.sparse_row) srow.push_back(row(v));
        for (const auto& v : s.sparse_col) scol.push_back(row(v));
        partial = s.partial;
        for (int i = 0; i < t; ++i)
            for (int j = 0; j < t; ++j) {
                mds[i][j] = s.mds[i][j];
                pre[i][j] = s.pre_sparse[i][j];
            }
        one_m = one();
    }
};

static inline void mat_vec9(const FastSpec& sp, const HFr (*m)[FastSpec::MAXT], HFr* s) {
    HFr out[FastSpec::MAXT];
    for (int i = 0; i < sp.t; ++i) {
        uint64_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < sp.t; ++j) wide_mac(acc, m[i][j], s[j]);
        out[i] = wide_redc(acc);
    }
    for (int i = 0; i < sp.t; ++i) s[i] = out[i];
}

// PoseidonChip::permutation (hash/poseidon.rs:193-230) in the crate's optimized schedule; `inputs` (n_in < T) are
// Montgomery values.  Same order of operations as oracle/poseidon.py::permutation.
static inline void permute(const FastSpec& sp, HFr* s, const HFr* inputs, int n_in) {
    const int t = sp.t;
    {   // absorb_with_pre_constants (hash/poseidon.rs:45-86)
        const auto& pc = sp.start[0];
        s[0] = fadd(s[0], pc[0]);
        for (int i = 0; i < n_in; ++i) s[i + 1] = fadd(fadd(s[i + 1], inputs[i]), pc[i + 1]);
        for (int i = n_in + 1; i < t; ++i) {
            s[i] = fadd(s[i], pc[i]);
            if (i == n_in + 1) s[i] = fadd(s[i], sp.one_m);
        }
    }
    for (int k = 1; k < sp.h; ++k) {
        for (int i = 0; i < t; ++i) s[i] = fadd(pow5(s[i]), sp.start[k][i]);
        mat_vec9(sp, sp.mds, s);
    }
    for (int i = 0; i < t; ++i) s[i] = fadd(pow5(s[i]), sp.start[sp.h][i]);
    mat_vec9(sp, sp.pre, s);
    for (int k = 0; k < sp.r_p; ++k) {
        const HFr s0 = fadd(pow5(s[0]), sp.partial[k]);
        uint64_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        wide_mac(acc, sp.srow[k][0], s0);
        for (int j = 1; j < t; ++j) wide_mac(acc, sp.srow[k][j], s[j]);
        for (int i = 1; i < t; ++i) {   // s_i = col_i * s0' + s_i, one reduction
            uint64_t a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            wide_mac(a2, sp.scol[k][i - 1], s0);
            wide_add_shifted(a2, s[i]);
            s[i] = wide_redc(a2);
        }
        s[0] = wide_redc(acc);
    }
    for (size_t k = 0; k < sp.end.size(); ++k) {
        for (int i = 0; i < t; ++i) s[i] = fadd(pow5(s[i]), sp.end[k][i]);
        mat_vec9(sp, sp.mds, s);
    }
    for (int i = 0; i < t; ++i) s[i] = pow5(s[i]);
    mat_vec9(sp, sp.mds, s);
}

// One sponge over a stream of canonical 32-byte elements with squeeze positions `upto` (non-decreasing): the host twin of
// k_poseidon_transcript.  PoseidonChip::squeeze (poseidon.rs:171-191): pending elements in chunks of RATE, one more
// permutation of the empty chunk when the last chunk was full or nothing was pending.  Returns false if an element is >= r.
static inline bool sponge_run(const FastSpec& sp, const uint8_t* elems, const uint32_t* upto, uint32_t nsq, uint8_t* out) {
    const int rate = sp.t - 1;
    HFr s[FastSpec::MAXT];
    for (int i = 0; i < sp.t; ++i) s[i] = zero();
    {
        const uint64_t w[4] = {0, 1, 0, 0};   // poseidon::State::default(): (2^64, 0, ..., 0)
        s[0] = from_words(w);
    }
    bool canonical = true;
    uint32_t pos = 0;
    for (uint32_t q = 0; q < nsq; ++q) {
        const uint32_t end = upto[q];
        uint32_t padding_offset = 0;
        bool any = false;
        while (pos < end) {
            const int nin = (int)((end - pos) < (uint32_t)rate ? (end - pos) : (uint32_t)rate);
            HFr in[FastSpec::MAXT];
            for (int i = 0; i < nin; ++i) {
                uint64_t w[4];
                memcpy(w, elems + 32 * (size_t)(pos + i), 32);
                if (geq_mod(w)) canonical = false;
                in[i] = from_words(w);
            }
            permute(sp, s, in, nin);
            padding_offset = (uint32_t)(rate - nin);
            pos += (uint32_t)nin;
            any = true;
        }
        if (!any || padding_offset == 0) permute(sp, s, nullptr, 0);
        uint64_t w[4];
        to_canonical(s[1], w);
        memcpy(out + 32 * (size_t)q, w, 32);
    }
    return canonical;
}

}  // namespace poseidon_host
}  // namespace h2agg
