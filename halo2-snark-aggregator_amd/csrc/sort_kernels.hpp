// Bucket sort of the MSM's (window, |digit|) keys for gfx950 — LDS atomics only on the hot path.
//
// Round-1 measurement (profiles/r01_kernel_stats_baseline_u32x8.txt): building the per-bucket runs with
// one device-scope atomic per (scalar, window) cost 0.63 ms (histogram) + 1.52 ms (scatter) for
// 2^20 x 16 keys — device-scope atomics leave the XCD (the 8 L2s are not coherent) and returning ones
// pay the round trip.  This version is a two-level partition:
//
//   level 1  a workgroup takes a tile of scalars, recodes them (registers), and counts its keys per
//            PARTITION = (window, high bits of the bucket) in LDS; one global atomic per (workgroup,
//            partition) reserves a contiguous slice, then the keys are written as (index|sign, low bits)
//            items — runs of ~TILE*W/PW items per slice instead of single scattered words
//   level 2  one workgroup per partition: LDS histogram over its <= 512 buckets, LDS scan -> hist[] /
//            offs[] of those buckets, then the items are placed into entries[] through LDS cursors; a
//            partition's slice is a few tens of KB, so the second read and the scattered 4-byte writes
//            stay in that XCD's L2
//   order    buckets are counting-sorted by length (descending) so the 64 lanes of a wave in the
//            accumulate kernel walk runs of (nearly) equal length
#pragma once
#include "batch_kernels.hpp"

namespace h2agg {

// The sort's kernels are latency chains (address arithmetic, loads, one returning LDS atomic per key); whatever shares their
// SIMDs — the previous MSM's Horner tail and window sums on the tail streams — is VALU-bound.  Raising the issue priority of
// the sort's waves serves the chains first: 1.281 -> 1.258 ms per 2^20-point step (profiles/r03_sweeps.txt; A/B build with
// -DH2AGG_NO_SORT_SETPRIO).
#ifndef H2AGG_NO_SORT_SETPRIO
#define SORT_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define SORT_PRIO() ((void)0)
#endif

constexpr int SORT_MAX_PW = 2048;   // partitions (LDS counters in level 1)
constexpr int SORT_SUB_BITS = 9;    // default low bucket bits resolved in level 2 (tunable, <= 12; sweep: profiles/r01_sweeps.txt)
constexpr int SORT_MAX_SUB_BITS = 12;
constexpr int SORT_MAX_SB = 1 << SORT_MAX_SUB_BITS;
constexpr int SIZE_BINS = 1024;     // bucket-length bins of the ordering pass

struct SortPlan {
    int sub_bits;      // min(9, c-1)
    uint32_t SB;       // buckets per partition
    uint32_t ppw;      // partitions per window = NB >> sub_bits
    uint32_t PW;       // partitions = W * ppw (<= SORT_MAX_PW)
    uint32_t tile;     // scalars per level-1 workgroup
    bool glv;          // `scalars` are glv_decompose() words
    // batched MSMs over ONE base table: scalar j belongs to MSM q = j / n_base, refers to base j - q * n_base and
    // its windows are numbered q * W1 + w (so every stage downstream just sees more windows).  n_base = 0: no batch.
    uint32_t n_base = 0;
    uint32_t W1 = 0;
    // fixed-base mode (Table::pre): the table holds 2^(c*w) * P_i for every digit position w at index w * pre_n + i, so
    // ALL digits of a scalar go to ONE bucket set (window q of the batch, or window 0) and refer to base w * pre_n + i
    uint32_t pre_n = 0;
};

// ------------------------------------------------------------------ GLV (endomorphism) decomposition
// BN254 G1 has the endomorphism phi(x, y) = (beta*x, y) = lambda*(x, y).  A scalar k splits as
// k = k1 + lambda*k2 (mod r) with |k1|, |k2| < 2^127, so s*P = k1*P + k2*phi(P): twice the points, half the
// windows — the bucket additions stay 16 per point at c = 16, but the bucket reduction and the serial Horner
// tail (c doublings per window) are halved.  Any integers (c1, c2) give an exact congruence; the rounding
// constants below only control the size: c_i is within 0.6 of the real solution, so
// |k1| <= 0.6 (a1 + a2) < 2^126.4 and |k2| <= 0.6 (|b1| + b2) < 2^126.4  (lattice basis from the extended
// Euclid on (r, lambda); a1 + b1*lambda = a2 + b2*lambda = 0 mod r; b1 < 0).  The kernel still checks the
// 127-bit bound and raises FLAG_NONCANONICAL if it is ever violated.
//   lambda = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
//   beta   = 0x30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd48
struct GlvConst {
    static constexpr uint32_t A1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
    static constexpr uint32_t B1N[2] = {0x94d213e3u, 0x89d32568u};   // -b1
    static constexpr uint32_t A2[2] = {0x94d213e3u, 0x89d32568u};
    static constexpr uint32_t B2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    static constexpr uint32_t G1[5] = {0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x00000002u};  // round(2^256 b2 / r)
    static constexpr uint32_t G2[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};                            // round(2^256 |b1| / r)
    static constexpr uint32_t BETA_MONT[9] = {0x18ccb791u, 0x175b1c3au, 0x0b83d6e2u, 0x0e8ed071u, 0x1282bee2u,
                                              0x04220e84u, 0x1fe4017fu, 0x15084d4au, 0x00169119u};
};

// out[0 .. NA+NB) = a * b  (little-endian 32-bit words, schoolbook with 64-bit column sums)
template <int NA, int NB>
FP_INLINE void mw_mul(const uint32_t (&a)[NA], const uint32_t (&b)[NB], uint32_t (&out)[NA + NB]) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NA + NB; ++k) {
        uint64_t lo = carry & 0xffffffffu, hi = carry >> 32;  // carry can exceed 32 bits: keep it as two halves
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int j = k - i;
            if (j >= 0 && j < NB) {
                const uint64_t pr = (uint64_t)a[i] * b[j];
                lo += pr & 0xffffffffu;
                hi += pr >> 32;
            }
        }
        out[k] = (uint32_t)lo;
        carry = hi + (lo >> 32);
    }
}
// r -= x (8 words, wraps mod 2^256); x given with NX <= 8 words
template <int NX>
FP_INLINE void mw_sub8(uint32_t (&r)[8], const uint32_t (&x)[NX]) {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t d = (uint64_t)r[i] - (i < NX ? x[i] : 0u) - br;
        r[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
}
// canonical k -> 256-bit word: low 128 = |k1| | sign1 << 127, high 128 = |k2| | sign2 << 127.  Returns false
// if a magnitude does not fit 127 bits (cannot happen for k < r, see above).
FP_INLINE bool glv_decompose(const U256& k, U256& d) {
    // c1 = (k*G1 + 2^255) >> 256, c2 = (k*G2 + 2^255) >> 256
    uint32_t t1[13], t2[11];
    mw_mul<8, 5>(k.w, GlvConst::G1, t1);
    mw_mul<8, 3>(k.w, GlvConst::G2, t2);
    uint32_t c1[4], c2[2];
    {
        uint64_t cy = (uint64_t)(t1[7] >> 31);  // + 2^255 rounding: carry into word 8 iff bit 255 is set
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cy += t1[8 + i];
            c1[i] = (uint32_t)cy;
            cy >>= 32;
        }
        cy = (uint64_t)(t2[7] >> 31);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            cy += t2[8 + i];
            c2[i] = (uint32_t)cy;
            cy >>= 32;
        }
    }
    // k1 = k - c1*a1 - c2*a2      k2 = c1*|b1| - c2*b2     (two's complement, 256 bits)
    uint32_t p11[8], p22[4], q1[6], q2[6];
    mw_mul<4, 4>(c1, GlvConst::A1, p11);
    mw_mul<2, 2>(c2, GlvConst::A2, p22);
    mw_mul<4, 2>(c1, GlvConst::B1N, q1);
    mw_mul<2, 4>(c2, GlvConst::B2, q2);
    uint32_t k1[8], k2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        k1[i] = k.w[i];
        k2[i] = i < 6 ? q1[i] : 0u;
    }
    mw_sub8<8>(k1, p11);
    mw_sub8<4>(k1, p22);
    mw_sub8<6>(k2, q2);
    bool ok = true;
    uint32_t* halves[2] = {k1, k2};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t* v = halves[h];
        const uint32_t neg = v[7] >> 31;
        if (neg) {  // magnitude = -v
            uint64_t cy = 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cy += (uint32_t)~v[i];
                v[i] = (uint32_t)cy;
                cy >>= 32;
            }
        }
        ok = ok && ((v[4] | v[5] | v[6] | v[7]) == 0) && ((v[3] >> 31) == 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) d.w[4 * h + i] = v[i];
        d.w[4 * h + 3] |= neg << 31;
    }
    return ok;
}

__global__ void __launch_bounds__(BLOCK) k_glv_decompose(const uint8_t* __restrict__ scalars, size_t n,
                                                         uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        U256 k = u256_load(scalars + 32 * i);
        uint32_t bad = !u256_is_canonical_fr(k);
        U256 d;
        bad |= !glv_decompose(k, d);
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        uint4* q = reinterpret_cast<uint4*>(out + 32 * i);
        q[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
        q[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
    }
}

// Signed-digit recode; calls f(window, bucket_index(0-based), negative, endo) for every non-zero digit.
// Digits lie in [-2^(c-1), 2^(c-1)].  glv = false: `s` is the canonical 256-bit scalar, W*c >= 255 guarantees
// no carry out of the top window.  glv = true: `s` is a glv_decompose() word — two sign-magnitude 127-bit
// halves, each recoded over the same W windows (W*c >= 128), the second half flagged `endo`.
template <class F>
FP_INLINE void msm_for_each_digit(U256 s, int c, int W, bool glv, F&& f) {
    const uint32_t mask = (1u << c) - 1u;
    const uint32_t half = 1u << (c - 1);
    if (!glv) {
        uint32_t carry = 0;
#pragma unroll 1
        for (int w = 0; w < W; ++w) {
            uint32_t raw = (s.w[0] & mask) + carry;
#pragma unroll
            for (int i = 0; i < 7; ++i) s.w[i] = (s.w[i] >> c) | (s.w[i + 1] << (32 - c));
            s.w[7] >>= c;
            const bool neg = raw > half;
            carry = neg ? 1u : 0u;
            const uint32_t mag = neg ? ((1u << c) - raw) : raw;
            if (mag != 0) f(w, mag - 1u, neg, false);
        }
        return;
    }
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        uint32_t m0 = s.w[4 * h], m1 = s.w[4 * h + 1], m2 = s.w[4 * h + 2], m3 = s.w[4 * h + 3];
        const bool sgn = (m3 >> 31) != 0;
        m3 &= 0x7fffffffu;
        uint32_t carry = 0;
#pragma unroll 1
        for (int w = 0; w < W; ++w) {
            uint32_t raw = (m0 & mask) + carry;
            m0 = (m0 >> c) | (m1 << (32 - c));
            m1 = (m1 >> c) | (m2 << (32 - c));
            m2 = (m2 >> c) | (m3 << (32 - c));
            m3 >>= c;
            const bool neg = raw > half;
            carry = neg ? 1u : 0u;
            const uint32_t mag = neg ? ((1u << c) - raw) : raw;
            if (mag != 0) f(w, mag - 1u, neg != sgn, h != 0);
        }
    }
}

// The same recoding, four digits at a time: `emit4(w0, nd, bkt[4], neg[4], ok[4], endo)` is called with up to four
// consecutive windows' digits so that the caller can put FOUR LDS atomics (and then four dependent stores) in flight
// instead of one: with `#pragma unroll 1` over single digits every returning atomic's latency was exposed, and a level-1
// workgroup (2 waves per SIMD at tile = 2048) has nobody to hide it behind.
template <class F>
FP_INLINE void msm_for_each_digit4(U256 s, int c, int W, bool glv, F&& emit4) {
    const uint32_t mask = (1u << c) - 1u;
    const uint32_t half = 1u << (c - 1);
    const int nh = glv ? 2 : 1;
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
        uint32_t m[8];
        bool sgn = false;
        if (glv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = s.w[4 * h + i];
#pragma unroll
            for (int i = 4; i < 8; ++i) m[i] = 0;
            sgn = (m[3] >> 31) != 0;
            m[3] &= 0x7fffffffu;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = s.w[i];
        }
        uint32_t carry = 0;
#pragma unroll 1
        for (int w0 = 0; w0 < W; w0 += 4) {
            uint32_t bkt[4];
            bool neg[4], ok[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t raw = (m[0] & mask) + carry;
                if (glv) {
                    m[0] = (m[0] >> c) | (m[1] << (32 - c));
                    m[1] = (m[1] >> c) | (m[2] << (32 - c));
                    m[2] = (m[2] >> c) | (m[3] << (32 - c));
                    m[3] >>= c;
                } else {
#pragma unroll
                    for (int i = 0; i < 7; ++i) m[i] = (m[i] >> c) | (m[i + 1] << (32 - c));
                    m[7] >>= c;
                }
                const bool ng = raw > half;
                const bool in = w0 + j < W;      // past the top window everything is zero: raw = carry = 0
                carry = (ng && in) ? 1u : 0u;
                const uint32_t mag = ng ? ((1u << c) - raw) : raw;
                ok[j] = in && mag != 0;
                bkt[j] = mag - 1u;
                neg[j] = ng != sgn;
            }
            emit4(w0, bkt, neg, ok, h != 0);
        }
    }
}

// packed level-1 sort item (32 bits): [ sub-bucket | neg | endo (only when glv) | idx : idx_bits ]
FP_INLINE uint32_t pack_item(uint32_t sub, bool neg, bool endo, uint32_t idx, int idx_bits, bool glv) {
    const int fb = glv ? 2 : 1;
    return (sub << (idx_bits + fb)) | ((neg ? 1u : 0u) << (idx_bits + fb - 1)) | ((glv && endo ? 1u : 0u) << idx_bits) | idx;
}

// entry / item bit layout of the point reference
constexpr uint32_t ENT_NEG = 0x80000000u;    // subtract the point
constexpr uint32_t ENT_ENDO = 0x40000000u;   // use phi(point) = (beta*x, y)
constexpr uint32_t ENT_IDX = 0x3fffffffu;

// Walk this thread's scalars of the tile (index i, value s) with the NEXT scalar's two 16-byte loads already
// in flight while the current one is recoded: the loop is otherwise one exposed memory latency per scalar.
#define TILE_SCALARS_BEGIN(TID)                                                        \
    {                                                                                  \
        uint32_t k_ = (TID);                                                           \
        size_t i = base + k_;                                                          \
        bool live_ = k_ < sp.tile && i < n;                                            \
        U256 nxt_;                                                                     \
        if (live_) nxt_ = u256_load(scalars + 32 * i);                                 \
        while (live_) {                                                                \
            U256 s = nxt_;                                                             \
            const uint32_t q_ = sp.n_base ? (uint32_t)i / sp.n_base : 0u;              \
            const uint32_t wofs = q_ * sp.W1;            /* window offset of MSM q */   \
            const uint32_t bi = (uint32_t)i - q_ * sp.n_base;   /* base index */        \
            /* window of digit w in the key space, base referred to by digit w */       \
            auto WIN = [&](int w_) { return sp.pre_n ? q_ : (uint32_t)w_ + wofs; };     \
            auto BASE = [&](int w_) { return sp.pre_n ? bi + (uint32_t)w_ * sp.pre_n : bi; }; \
            (void)BASE;                                                                 \
            const uint32_t kn_ = k_ + BLOCK;                                           \
            const size_t in_ = base + kn_;                                             \
            const bool more_ = kn_ < sp.tile && in_ < n;                               \
            if (more_) nxt_ = u256_load(scalars + 32 * in_);
#define TILE_SCALARS_END                                                               \
            k_ = kn_;                                                                  \
            i = in_;                                                                   \
            live_ = more_;                                                             \
        }                                                                              \
    }

// ------------------------------------------------------------------ level 1
// tile_counts (optional): this tile's count of every partition, [tile][PW] — the scatter pass of the same tile reads it
// back instead of recoding its scalars a second time just to count
__global__ void __launch_bounds__(BLOCK) k_part_count(const uint8_t* __restrict__ scalars, size_t n, int c, int W,
                                                      SortPlan sp, uint32_t* __restrict__ pcount, uint32_t* flags,
                                                      uint32_t* __restrict__ tile_counts) {
    __shared__ uint32_t cnt[SORT_MAX_PW];
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) cnt[p] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * sp.tile;
    uint32_t bad = 0;
    TILE_SCALARS_BEGIN(threadIdx.x)
        if (!sp.glv) bad |= !u256_is_canonical_fr(s);  // (GLV words were range-checked by k_glv_decompose)
        msm_for_each_digit(s, c, W, sp.glv, [&](int w, uint32_t b, bool, bool) {
            atomicAdd(&cnt[WIN(w) * sp.ppw + (b >> sp.sub_bits)], 1u);
        });
    TILE_SCALARS_END
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) {
        const uint32_t v = cnt[p];
        if (v) atomicAdd(&pcount[p], v);
        if (tile_counts) tile_counts[(size_t)blockIdx.x * sp.PW + p] = v;
    }
}

// exclusive scan of up to 2 * SORT_MAX_PW values by one workgroup: out[i] = sum_{j<i} in[j], out[n] = total;
// cursor (optional) receives a copy of out[0..n)
__global__ void __launch_bounds__(BLOCK) k_scan_small(const uint32_t* __restrict__ in, uint32_t n,
                                                      uint32_t* __restrict__ out, uint32_t* __restrict__ cursor) {
    SORT_PRIO();
    __shared__ uint32_t lds[BLOCK];
    constexpr int PER = 2 * SORT_MAX_PW / BLOCK;  // 8
    const uint32_t base = threadIdx.x * PER;
    uint32_t v[PER];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    const int tid = threadIdx.x;
    lds[tid] = sum;
    __syncthreads();
#pragma unroll 1
    for (int d = 1; d < BLOCK; d <<= 1) {
        uint32_t t = (tid >= d) ? lds[tid - d] : 0u;
        __syncthreads();
        lds[tid] += t;
        __syncthreads();
    }
    uint32_t off = lds[tid] - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (base + k < n) {
            out[base + k] = off;
            if (cursor) cursor[base + k] = off;
        }
        off += v[k];
    }
    if (tid == BLOCK - 1) out[n] = lds[BLOCK - 1];
}

__global__ void __launch_bounds__(BLOCK) k_part_scatter(const uint8_t* __restrict__ scalars, size_t n, int c, int W,
                                                        SortPlan sp, uint32_t* __restrict__ pcursor,
                                                        uint32_t* __restrict__ item_idx, uint16_t* __restrict__ item_sub) {
    __shared__ uint32_t cnt[SORT_MAX_PW];
    __shared__ uint32_t basep[SORT_MAX_PW];
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) cnt[p] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * sp.tile;
    TILE_SCALARS_BEGIN(threadIdx.x)
        msm_for_each_digit(s, c, W, sp.glv, [&](int w, uint32_t b, bool, bool) {
            atomicAdd(&cnt[WIN(w) * sp.ppw + (b >> sp.sub_bits)], 1u);
        });
    TILE_SCALARS_END
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) {
        const uint32_t v = cnt[p];
        basep[p] = v ? atomicAdd(&pcursor[p], v) : 0u;  // one device-scope atomic per (workgroup, partition)
        cnt[p] = 0;
    }
    __syncthreads();
    const uint32_t submask = sp.SB - 1u;
    TILE_SCALARS_BEGIN(threadIdx.x)
        msm_for_each_digit(s, c, W, sp.glv, [&](int w, uint32_t b, bool neg, bool endo) {
            const uint32_t p = WIN(w) * sp.ppw + (b >> sp.sub_bits);
            const uint32_t pos = basep[p] + atomicAdd(&cnt[p], 1u);
            item_idx[pos] = BASE(w) | (neg ? ENT_NEG : 0u) | (endo ? ENT_ENDO : 0u);
            item_sub[pos] = (uint16_t)(b & submask);
        });
    TILE_SCALARS_END
}

// ------------------------------------------------------------------ level 2: one workgroup per partition
__global__ void __launch_bounds__(BLOCK) k_bucket_sort(const uint32_t* __restrict__ pstart,
                                                       const uint32_t* __restrict__ item_idx,
                                                       const uint16_t* __restrict__ item_sub, SortPlan sp, uint32_t NB,
                                                       uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                       uint32_t* __restrict__ entries) {
    __shared__ uint32_t h[SORT_MAX_SB];
    __shared__ uint32_t scan[BLOCK];
    const uint32_t p = blockIdx.x;
    const uint32_t start = pstart[p], end = pstart[p + 1];
    const int tid = threadIdx.x;
    for (uint32_t s = tid; s < sp.SB; s += BLOCK) h[s] = 0;
    __syncthreads();
    for (uint32_t k0 = start + tid; k0 < end; k0 += 8 * BLOCK) {  // 8 independent loads in flight
        uint16_t sb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sb[j] = (k0 + j * BLOCK < end) ? item_sub[k0 + j * BLOCK] : (uint16_t)0xffff;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (sb[j] != 0xffff) atomicAdd(&h[sb[j]], 1u);
    }
    __syncthreads();
    // exclusive scan of h[0..SB): each thread owns `per` consecutive counters
    const uint32_t per = (sp.SB + BLOCK - 1) / BLOCK;
    const uint32_t lo = tid * per;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j)
        if (lo + j < sp.SB) mine += h[lo + j];
    scan[tid] = mine;
    __syncthreads();
#pragma unroll 1
    for (int d = 1; d < BLOCK; d <<= 1) {
        uint32_t t = (tid >= d) ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    uint32_t run = scan[tid] - mine;
    const uint32_t w = p / sp.ppw, phi = p - w * sp.ppw;
    const uint32_t key0 = w * NB + (phi << sp.sub_bits);
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t sidx = lo + j;
        if (sidx < sp.SB) {
            const uint32_t cnt = h[sidx];
            hist[key0 + sidx] = cnt;
            offs[key0 + sidx] = start + run;
            h[sidx] = run;  // becomes the cursor
            run += cnt;
        }
    }
    __syncthreads();
    for (uint32_t k0 = start + tid; k0 < end; k0 += 8 * BLOCK) {
        uint16_t sb[8];
        uint32_t ix[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = k0 + j * BLOCK < end;
            sb[j] = ok ? item_sub[k0 + j * BLOCK] : (uint16_t)0xffff;
            ix[j] = ok ? item_idx[k0 + j * BLOCK] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (sb[j] != 0xffff) entries[start + atomicAdd(&h[sb[j]], 1u)] = ix[j];
    }
}

// ------------------------------------------------------------------ LDS-staged variants (n < 2^idx_bits)
// Round-1 PMC (profiles/earlier/r01_pmc_write.txt): the direct versions above write 563 MB (level 1) and 426 MB
// (level 2) for 96 MB + 64 MB of payload — single 4- and 2-byte stores to ~1000 open runs per workgroup are
// evicted from L2 as partial lines.  Here the keys of a tile (level 1) / of a partition (level 2) are
// first ordered in LDS and then leave the CU as contiguous runs written by consecutive lanes.
// Item = sub-bucket | negative | endo (GLV only) | point index, packed from the top; see pack_item().
constexpr int STAGE_ITEMS = 32768;  // 128 KiB of LDS
constexpr int STAGE_ITEMS_L1 = 24576;  // level-1 staging shares the LDS with 4 x SORT_MAX_PW counters

__global__ void __launch_bounds__(BLOCK) k_part_scatter_staged(const uint8_t* __restrict__ scalars, size_t n, int c,
                                                               int W, SortPlan sp, int idx_bits,
                                                               uint32_t* __restrict__ pcursor,
                                                               uint32_t* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* len = smem;                       // [SORT_MAX_PW]
    uint32_t* lbase = smem + SORT_MAX_PW;       // [SORT_MAX_PW]
    uint32_t* gbase = smem + 2 * SORT_MAX_PW;   // [SORT_MAX_PW]
    uint32_t* cur = smem + 3 * SORT_MAX_PW;     // [SORT_MAX_PW]
    uint32_t* stage = smem + 4 * SORT_MAX_PW;   // [STAGE_ITEMS]
    const int tid = threadIdx.x;
    for (uint32_t p = tid; p < SORT_MAX_PW; p += BLOCK) {
        len[p] = 0;
        cur[p] = 0;
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * sp.tile;
    TILE_SCALARS_BEGIN(tid)
        msm_for_each_digit(s, c, W, sp.glv, [&](int w, uint32_t b, bool, bool) {
            atomicAdd(&len[WIN(w) * sp.ppw + (b >> sp.sub_bits)], 1u);
        });
    TILE_SCALARS_END
    __syncthreads();
    // exclusive scan over the partitions (4 per thread), and one global reservation per non-empty partition
    constexpr int PER = SORT_MAX_PW / BLOCK;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = len[tid * PER + j];
        sum += v[j];
    }
    gbase[tid] = sum;  // scratch for the scan
    __syncthreads();
#pragma unroll 1
    for (int d = 1; d < BLOCK; d <<= 1) {
        uint32_t t = (tid >= d) ? gbase[tid - d] : 0u;
        __syncthreads();
        gbase[tid] += t;
        __syncthreads();
    }
    uint32_t off = gbase[tid] - sum;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t p = tid * PER + j;
        lbase[p] = off;
        off += v[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t p = tid * PER + j;
        gbase[p] = v[j] ? atomicAdd(&pcursor[p], v[j]) : 0u;
    }
    __syncthreads();
    const uint32_t submask = sp.SB - 1u;
    TILE_SCALARS_BEGIN(tid)
        msm_for_each_digit(s, c, W, sp.glv, [&](int w, uint32_t b, bool neg, bool endo) {
            const uint32_t p = WIN(w) * sp.ppw + (b >> sp.sub_bits);
            const uint32_t r = atomicAdd(&cur[p], 1u);
            stage[lbase[p] + r] = pack_item(b & submask, neg, endo, BASE(w), idx_bits, sp.glv);
        });
    TILE_SCALARS_END
    __syncthreads();
    // copy-out: a wave per partition run, consecutive lanes -> consecutive addresses
    const uint32_t wave = tid >> 6, lane = tid & 63;
    for (uint32_t p = wave; p < sp.PW; p += BLOCK / 64) {
        const uint32_t L = len[p], lb = lbase[p], gb = gbase[p];
        for (uint32_t k = lane; k < L; k += 64) items[gb + k] = stage[lb + k];
    }
}

// direct level 1 writing PACKED items (small LDS footprint -> several workgroups per CU hide the latency
// that the 144-KiB staged variant exposes; profiles/r01_sweeps.txt)
__global__ void __launch_bounds__(BLOCK) k_part_scatter_packed(const uint8_t* __restrict__ scalars, size_t n, int c,
                                                               int W, SortPlan sp, int idx_bits,
                                                               uint32_t* __restrict__ pcursor,
                                                               uint32_t* __restrict__ items,
                                                               const uint32_t* __restrict__ tile_counts) {
    __shared__ uint32_t cnt[SORT_MAX_PW];
    __shared__ uint32_t basep[SORT_MAX_PW];
    const size_t base = (size_t)blockIdx.x * sp.tile;
    if (tile_counts) {   // counted by k_part_count over the same tile: one global reservation per non-empty partition
        for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) {
            const uint32_t v = tile_counts[(size_t)blockIdx.x * sp.PW + p];
            basep[p] = v ? atomicAdd(&pcursor[p], v) : 0u;
            cnt[p] = 0;
        }
    } else {
        for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) cnt[p] = 0;
        __syncthreads();
        TILE_SCALARS_BEGIN(threadIdx.x)
            msm_for_each_digit(s, c, W, sp.glv, [&](int w, uint32_t b, bool, bool) {
                atomicAdd(&cnt[WIN(w) * sp.ppw + (b >> sp.sub_bits)], 1u);
            });
        TILE_SCALARS_END
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) {
            const uint32_t v = cnt[p];
            basep[p] = v ? atomicAdd(&pcursor[p], v) : 0u;
            cnt[p] = 0;
        }
    }
    __syncthreads();
    const uint32_t submask = sp.SB - 1u;
    TILE_SCALARS_BEGIN(threadIdx.x)
        msm_for_each_digit4(s, c, W, sp.glv, [&](int w0, const uint32_t (&b)[4], const bool (&neg)[4], const bool (&ok)[4], bool endo) {
            uint32_t p[4], r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = ok[j] ? WIN(w0 + j) * sp.ppw + (b[j] >> sp.sub_bits) : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = ok[j] ? basep[p[j]] + atomicAdd(&cnt[p[j]], 1u) : 0u;   // four in flight
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ok[j]) items[r[j]] = pack_item(b[j] & submask, neg[j], endo, BASE(w0 + j), idx_bits, sp.glv);
        });
    TILE_SCALARS_END
}

// TB threads per workgroup: the 128-KiB stage allows ONE workgroup per CU, so its size is the CU's whole occupancy —
// 256 threads left one wave per SIMD to hide the item loads and the LDS atomics' round trips behind (round 2: 1024).
template <int TB>
__global__ void __launch_bounds__(TB) k_bucket_sort_staged(const uint32_t* __restrict__ pstart,
                                                              const uint32_t* __restrict__ items, SortPlan sp,
                                                              int idx_bits, uint32_t NB, uint32_t* __restrict__ hist,
                                                              uint32_t* __restrict__ offs,
                                                              uint32_t* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* h = smem;                         // [SORT_MAX_SB]
    uint32_t* scan = smem + SORT_MAX_SB;        // [TB]
    uint32_t* sorted = smem + SORT_MAX_SB + TB;  // [STAGE_ITEMS]
    const uint32_t p = blockIdx.x;
    const uint32_t start = pstart[p], end = pstart[p + 1], total = end - start;
    const int tid = threadIdx.x;
    const uint32_t idxmask = (1u << idx_bits) - 1u;
    const int fb = sp.glv ? 2 : 1;  // flag bits between the index and the sub-bucket: neg, and endo under GLV
    for (uint32_t s = tid; s < sp.SB; s += TB) h[s] = 0;
    __syncthreads();
    for (uint32_t k0 = start + tid; k0 < end; k0 += 8 * TB) {  // 8 independent loads in flight
        uint32_t it[8];  // every 32-bit pattern is a legal item (sub = max, negative, last index): no sentinel
#pragma unroll
        for (int j = 0; j < 8; ++j) it[j] = (k0 + j * TB < end) ? items[k0 + j * TB] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j * TB < end) atomicAdd(&h[it[j] >> (idx_bits + fb)], 1u);
    }
    __syncthreads();
    const uint32_t per = (sp.SB + TB - 1) / TB;
    const uint32_t lo = tid * per;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j)
        if (lo + j < sp.SB) mine += h[lo + j];
    scan[tid] = mine;
    __syncthreads();
#pragma unroll 1
    for (int d = 1; d < TB; d <<= 1) {
        uint32_t t = (tid >= d) ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    uint32_t run = scan[tid] - mine;
    const uint32_t w = p / sp.ppw, phi = p - w * sp.ppw;
    const uint32_t key0 = w * NB + (phi << sp.sub_bits);
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t sidx = lo + j;
        if (sidx < sp.SB) {
            const uint32_t cnt = h[sidx];
            hist[key0 + sidx] = cnt;
            offs[key0 + sidx] = start + run;
            h[sidx] = run;  // becomes the cursor
            run += cnt;
        }
    }
    __syncthreads();
    const bool staged = total <= (uint32_t)STAGE_ITEMS;
    for (uint32_t k0 = start + tid; k0 < end; k0 += 8 * TB) {
        uint32_t itv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) itv[j] = (k0 + j * TB < end) ? items[k0 + j * TB] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t it = itv[j];
            if (k0 + j * TB >= end) continue;
            const uint32_t r = atomicAdd(&h[it >> (idx_bits + fb)], 1u);
            const uint32_t e = (it & idxmask) | ((sp.glv ? ((it >> idx_bits) & 1u) : 0u) << 30) |
                               (((it >> (idx_bits + fb - 1)) & 1u) << 31);
            if (staged) sorted[r] = e;
            else entries[start + r] = e;  // over-long partition (skewed scalars): direct placement
        }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t k = tid; k < total; k += TB) entries[start + k] = sorted[k];
    }
}

// ------------------------------------------------------------------ small MSMs: the whole sort in ONE launch
// n <= SMALL_SORT_N scalars (the two multi_exps of an evaluation: ~350 and ~1 400 points at 4 proofs; short instance columns):
// the four launches of the packed path (count, scan, scatter, bucket sort) are 35-55 us of mostly launch latency for ~50 K keys,
// on the critical path of an evaluation that is 0.65 ms in all.  Here a workgroup owns ONE window of ONE MSM: it recodes its MSM's
// scalars (twice: count, place — 32 B each, out of L2), counts its window's digits in LDS, scans, and writes `entries`
// bucket-ordered into the window's fixed slice, plus hist / offs: exactly what k_bucket_sort* leave behind.
// Three shapes: one MSM (n_base = split = 0); a batch over one table (scalar j -> MSM j / n_base, base j % n_base); a SPLIT
// table — two MSMs over the disjoint parts [0, split) and [split, n) of one table, windows numbered q * W + w like a batch's
// (the evaluation's w_x and w_g sides as one set of launches).
constexpr int SMALL_SORT_N = 16384;
constexpr int SMALL_SORT_TB = 1024;
constexpr int SMALL_SORT_NB = 8192;    // buckets of a window kept in LDS (c <= 14)
__global__ void __launch_bounds__(SMALL_SORT_TB) k_small_sort(const uint8_t* __restrict__ scalars, uint32_t n, int c, int W,
                                                              uint32_t n_base, uint32_t split, bool glv, uint32_t NB,
                                                              uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                              uint32_t* __restrict__ entries, uint32_t* flags,
                                                              uint32_t* __restrict__ counters) {
    // the accumulation's list counters (over-long buckets, fix-ups: three words) start at zero; nothing else of the sort's
    // scratch words is read on this path, so the memset of all of them (~5 us in front of the sort) is not needed
    if (blockIdx.x == 0 && threadIdx.x < 4) counters[threadIdx.x] = 0;
    __shared__ uint32_t h[SMALL_SORT_NB];
    __shared__ uint32_t scan[SMALL_SORT_TB];
    constexpr uint32_t TB = SMALL_SORT_TB;
    const uint32_t tid = threadIdx.x;
    const uint32_t win = blockIdx.x, q = win / (uint32_t)W, w = win - q * (uint32_t)W;
    const uint32_t g = glv ? 2u : 1u;
    uint32_t lo, hi, bofs, start;   // this MSM's scalars [lo, hi), what to subtract for the base index, the window's slice of `entries`
    if (split) {
        lo = q ? split : 0u;
        hi = q ? n : split;
        bofs = 0;
        start = q ? ((uint32_t)W * split + w * (n - split)) * g : w * split * g;
    } else if (n_base) {
        lo = q * n_base;
        hi = lo + n_base;
        bofs = lo;
        start = win * n_base * g;
    } else {
        lo = 0;
        hi = n;
        bofs = 0;
        start = win * n * g;
    }
    for (uint32_t b = tid; b < NB; b += TB) h[b] = 0;
    __syncthreads();
    uint32_t bad = 0;
    for (uint32_t i = lo + tid; i < hi; i += TB) {
        const U256 s = u256_load(scalars + 32 * (size_t)i);
        if (!glv && w == 0) bad |= !u256_is_canonical_fr(s);   // (GLV words were range-checked by k_glv_decompose)
        msm_for_each_digit(s, c, W, glv, [&](int ww, uint32_t b, bool, bool) {
            if ((uint32_t)ww == w) atomicAdd(&h[b], 1u);
        });
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    __syncthreads();
    const uint32_t per = (NB + TB - 1) / TB;
    const uint32_t first = tid * per;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j)
        if (first + j < NB) mine += h[first + j];
    scan[tid] = mine;
    __syncthreads();
#pragma unroll 1
    for (uint32_t d = 1; d < TB; d <<= 1) {
        const uint32_t t = (tid >= d) ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    uint32_t run = scan[tid] - mine;
    const uint32_t key0 = win * NB;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t b = first + j;
        if (b < NB) {
            const uint32_t cnt = h[b];
            hist[key0 + b] = cnt;
            offs[key0 + b] = start + run;
            h[b] = run;   // becomes the cursor
            run += cnt;
        }
    }
    __syncthreads();
    for (uint32_t i = lo + tid; i < hi; i += TB) {
        const U256 s = u256_load(scalars + 32 * (size_t)i);
        const uint32_t bi = i - bofs;
        msm_for_each_digit(s, c, W, glv, [&](int ww, uint32_t b, bool neg, bool endo) {
            if ((uint32_t)ww == w) {
                const uint32_t r = atomicAdd(&h[b], 1u);
                entries[start + r] = bi | (endo ? ENT_ENDO : 0u) | (neg ? ENT_NEG : 0u);
            }
        });
    }
}

// ------------------------------------------------------------------ digit-major sort (plain MSM, c = 16, n <= 2^22)
// Round-2 PMC (profiles/r02_final_pmc_*.txt): the packed level 1 above is bound by its 16.7 M scattered 4-byte stores (one L2
// request each, 278 MB written for 64 MB of payload) and by occupancy (a thread recodes a whole 256-bit scalar per digit walk,
// 620 instructions per scalar, twice), not by LDS or HBM.  This path turns the problem by 90 degrees:
//   digits     one pass writes the signed digits WINDOW-MAJOR as 16-bit codes (32 MiB at 2^20 points): everything after
//              it reads 2 coalesced bytes per key and never touches a scalar again
//   level 1    a workgroup takes (window, tile of 8192 points): 64..512 partitions of ONE window, so a partition's run in a
//              tile is ~128 keys; the tile is ordered in 32 KiB of LDS and written back IN PLACE (tile-local, coalesced) with
//              its table of partition offsets — no global reservation, no second recoding pass; the rank a key got from its
//              one returning LDS atomic is kept in a register and reused for the placement
//   level 2    one workgroup per partition collects its ~128-key run from every tile (contiguous, offsets from the tables),
//              and sorts by bucket in LDS with ONE atomic per key as well (rank kept in registers)
// LDS atomics per key: 2 (4 before); scattered global stores: none.
constexpr int DM_T1 = 8192;          // keys per level-1 workgroup
constexpr int DM_TB1 = 512;          // its threads
constexpr int DM_PER1 = DM_T1 / DM_TB1;
constexpr int DM_MAX_PPW = 512;      // partitions per window
constexpr int DM_MAX_TILES = 512;    // n <= 2^22
constexpr int DM_TB2 = 512;          // level-2 workgroup
// keys a level-2 workgroup orders in LDS: PER per thread, 8192 (32 KiB, mean partition 4096) or 16384 (2^22 points: mean 8192).
// The keys and their ranks sit in registers between the passes: PER = 16 keeps the kernel under 96 VGPRs, so that two workgroups
// fit a CU NEXT TO the previous MSM's bucket reduction (128 VGPRs per wave), which runs on the tail stream during the sort.
constexpr int DM_MAX_PW = 16 * DM_MAX_PPW;
constexpr uint32_t DM_ZERO = 0x8000u;   // code of a zero digit; magnitude m > 0: m - 1 (positive), 0x8000 | m (negative)

struct DmPlan {
    uint32_t n, n_pad;     // points; row stride of codes[] (multiple of 8)
    uint32_t n_row;        // row stride of items[]: ntile * DM_T1 (whole tiles: a tile's vector stores may run 3 keys past its last one)
    uint32_t ntile;        // level-1 tiles per window
    int sub_bits;          // low bucket bits resolved in level 2
    uint32_t SB, ppw;      // buckets per partition, partitions per window
    int idx_bits;          // 31 - sub_bits: item = sub | neg | idx
    uint32_t n_pts;        // GLV: a row holds the keys of both halves, idx >= n_pts = second half (endo) of point idx - n_pts;
                           // otherwise 0xffffffff
};

// codes[w * n_pad + i] = signed 16-bit digit w of scalar i (c = 16, W = 16; r < 2^254: no carry out of the top window)
constexpr int DM_DIG_PER = 8;   // scalars per thread: few, long-lived waves (wave launches crawl while a VALU-saturating
                                // kernel of the previous MSM's tail shares the SIMDs: 113 us instead of 20 with one scalar each)
__global__ void __launch_bounds__(BLOCK) k_dm_digits(const uint8_t* __restrict__ scalars, uint32_t n, uint32_t n_pad,
                                                     uint16_t* __restrict__ codes, uint32_t* flags) {
    SORT_PRIO();
    const uint32_t i0 = blockIdx.x * (BLOCK * DM_DIG_PER) + threadIdx.x;
    U256 nxt;
    if (i0 < n) nxt = u256_load(scalars + 32 * (size_t)i0);
    uint32_t bad = 0;
#pragma unroll 1
    for (int k = 0; k < DM_DIG_PER; ++k) {
        const uint32_t i = i0 + k * BLOCK;
        if (i >= n) break;
        const U256 s = nxt;
        if (k + 1 < DM_DIG_PER && i + BLOCK < n) nxt = u256_load(scalars + 32 * (size_t)(i + BLOCK));
        bad |= !u256_is_canonical_fr(s);
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t raw = ((s.w[w >> 1] >> (16 * (w & 1))) & 0xffffu) + carry;
            const bool neg = raw > 0x8000u;
            carry = neg ? 1u : 0u;
            const uint32_t code = neg ? (0x8000u | (0x10000u - raw)) : (raw == 0 ? DM_ZERO : raw - 1u);
            codes[(size_t)w * n_pad + i] = (uint16_t)code;
        }
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
}

// c = 17 (15 windows, 2^16 buckets each): the magnitudes m in [1, 2^16] take all 65 536 16-bit codes (m - 1), so sign and
// "zero digit" travel beside them as BIT rows (one bit per key, written 64 at a time from a wave's ballot):
//   codes[w * n_pad + i] = m - 1 (0 for a zero digit)      sign row w, zero row w: n_pad / 8 bytes each, behind the codes
// — 2.25 bytes per key instead of 2, one window less everywhere else.  Window w covers bits [17 w, 17 w + 17); r < 2^254 leaves
// the top window (w = 14, bits 238..253) 16 bits + the carry: no carry out.  n_pad is a multiple of 64 here.
constexpr int DM17_W = 15;
FP_INLINE const uint8_t* dm17_flags(const uint16_t* codes, uint32_t n_pad) {
    return reinterpret_cast<const uint8_t*>(codes + (size_t)DM17_W * n_pad);
}
__global__ void __launch_bounds__(BLOCK) k_dm_digits17(const uint8_t* __restrict__ scalars, uint32_t n, uint32_t n_pad,
                                                       uint16_t* __restrict__ codes, uint32_t* flags) {
    SORT_PRIO();
    const uint32_t i0 = blockIdx.x * (BLOCK * DM_DIG_PER) + threadIdx.x;
    uint64_t* bits = reinterpret_cast<uint64_t*>(codes + (size_t)DM17_W * n_pad);   // [2 * DM17_W][n_pad / 64]
    const uint32_t row64 = n_pad >> 6;
    uint32_t bad = 0;
#pragma unroll 1
    for (int k = 0; k < DM_DIG_PER; ++k) {
        const uint32_t i = i0 + k * BLOCK;
        const bool act = i < n;
        if (__ballot(i < n_pad) == 0) break;            // (wave-uniform: the wave's 64 keys start at a multiple of 64)
        U256 s;
#pragma unroll
        for (int j = 0; j < 8; ++j) s.w[j] = 0;
        if (act) s = u256_load(scalars + 32 * (size_t)i);
        bad |= act && !u256_is_canonical_fr(s);
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < DM17_W; ++w) {
            const int pos = 17 * w, wd = pos >> 5, sh = pos & 31;
            uint32_t v = s.w[wd] >> sh;
            if (sh > 15 && wd + 1 < 8) v |= s.w[wd + 1] << (32 - sh);
            const uint32_t raw = (v & 0x1ffffu) + carry;
            const bool neg = raw > 0x10000u;
            carry = neg ? 1u : 0u;
            const uint32_t m = neg ? 0x20000u - raw : raw;   // 0 .. 2^16
            const uint64_t sm = __ballot(act && neg), zm = __ballot(!act || m == 0);
            if (i < n_pad) codes[(size_t)w * n_pad + i] = (uint16_t)(m ? m - 1u : 0u);
            if ((threadIdx.x & 63) == 0 && i < n_pad) {
                bits[(size_t)w * row64 + (i >> 6)] = sm;
                bits[(size_t)(DM17_W + w) * row64 + (i >> 6)] = zm;
            }
        }
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
}

// GLV: `words` are glv_decompose() outputs (two sign-magnitude 127-bit halves).  Both halves are recoded over the same 8
// windows; row w holds the first halves' digits at [0, n) and the second halves' at [n, 2n).
__global__ void __launch_bounds__(BLOCK) k_dm_digits_glv(const uint8_t* __restrict__ words, uint32_t n, uint32_t n_pad,
                                                         uint16_t* __restrict__ codes) {
    SORT_PRIO();
    const uint32_t i0 = blockIdx.x * (BLOCK * DM_DIG_PER) + threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < DM_DIG_PER; ++k) {
        const uint32_t i = i0 + k * BLOCK;
        if (i >= n) break;
        const U256 s = u256_load(words + 32 * (size_t)i);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t sgn = s.w[4 * h + 3] >> 31;
            uint32_t carry = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                uint32_t hw = (s.w[4 * h + (w >> 1)] >> (16 * (w & 1))) & 0xffffu;
                if (w == 7) hw &= 0x7fffu;   // bit 127 is the sign
                const uint32_t raw = hw + carry;
                // digits of a half lie in [-(2^15 - 1), 2^15]; a negative half flips them to [-2^15, 2^15 - 1], and -2^15
                // has no 16-bit code.  Recoding a negative half with the mirrored threshold (raw 2^15 -> digit -2^15, carry)
                // keeps its flipped digits in [-(2^15 - 1), 2^15] as well.  |k_i| < 2^126.4: no carry out of window 7.
                const bool big = raw > (sgn ? 0x7fffu : 0x8000u);
                carry = big ? 1u : 0u;
                const uint32_t mag = big ? 0x10000u - raw : raw;            // 0 .. 0x8000
                const uint32_t neg = (big ? 1u : 0u) ^ sgn;
                const uint32_t code = mag == 0 ? DM_ZERO : (neg ? (0x8000u | (mag & 0x7fffu)) : mag - 1u);
                codes[(size_t)w * n_pad + (size_t)h * n + i] = (uint16_t)code;
            }
        }
    }
}

// exclusive scan of v[0..cnt) (cnt <= 512) in LDS, total -> v[cnt]: ONE wave does it (8 values per lane + a 6-step wave
// scan), everybody else waits at the barrier — a Hillis-Steele scan over a 1024-thread workgroup is 20 barriers of 16 waves
FP_INLINE uint32_t dm_wave_scan_incl(uint32_t x, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += t;
    }
    return x;
}
FP_INLINE void dm_block_scan(uint32_t* v, uint32_t cnt) {
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        uint32_t x[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = (8 * lane + j < cnt) ? v[8 * lane + j] : 0u;
            sum += x[j];
        }
        const uint32_t inc = dm_wave_scan_incl(sum, lane);
        uint32_t off = inc - sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (8 * lane + j < cnt) v[8 * lane + j] = off;
            off += x[j];
        }
        if (lane == 63) v[cnt] = inc;
    }
    __syncthreads();
}

// out[i] = sum_{j<i} in[j] for i <= n (n <= DM_MAX_PW = 8192): one 1024-thread workgroup, 8 values per thread, wave scans
__global__ void __launch_bounds__(1024) k_dm_scan(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    SORT_PRIO();
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t x[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x[j] = (8 * tid + j < n) ? in[8 * tid + j] : 0u;
        sum += x[j];
    }
    const uint32_t inc = dm_wave_scan_incl(sum, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wave; ++k) base += wsum[k];
    uint32_t off = base + inc - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (8 * tid + j <= n) out[8 * tid + j] = off;
        off += x[j];
    }
    if (8 * tid + 8 == n) out[n] = off;   // n = 8 * blockDim.x: the total has no thread of its own
}

// grid (ntile, W).  items[w * n_row + tile * DM_T1 + k], k < toff[..][ppw]: the tile's keys ordered by partition;
// toff[(w * ntile + tile) * (ppw + 1) + p]: where partition p starts inside the tile; pcount[w * ppw + p] += its length
template <bool WIDE>   // WIDE: the 17-bit windows of k_dm_digits17 — 16-bit magnitude codes as usual plus one sign and one zero bit per key in bit rows behind them; otherwise the plain 16-bit codes
__global__ void __launch_bounds__(DM_TB1) k_dm_partition(const uint16_t* __restrict__ codes, DmPlan dp,
                                                         uint32_t* __restrict__ pcount, uint32_t* __restrict__ toff,
                                                         uint32_t* __restrict__ items) {
    SORT_PRIO();
    __shared__ uint32_t cnt[DM_MAX_PPW + 1];
    __shared__ __attribute__((aligned(16))) uint32_t stage[DM_T1];
    const uint32_t tile = blockIdx.x, w = blockIdx.y;
    const int tid = threadIdx.x;
    for (uint32_t p = tid; p <= dp.ppw; p += DM_TB1) cnt[p] = 0;
    __syncthreads();
    const size_t row = (size_t)w * dp.n_pad;
    const uint32_t t0 = tile * DM_T1;
    // a thread's DM_PER1 keys sit in runs of G = 8 consecutive points: one 16-byte load of codes each (WIDE: + the run's sign
    // and zero bytes of k_dm_digits17's bit rows)
    constexpr int G = 8, NV = DM_PER1 / G;
    uint4 v[NV];
    uint32_t sg[NV], zr[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const uint32_t i0 = t0 + (k * DM_TB1 + tid) * G;
        sg[k] = 0;
        zr[k] = 0xffu;
        if (WIDE) {
            v[k] = make_uint4(0, 0, 0, 0);
            if (i0 < dp.n_pad) {
                const uint8_t* fl = dm17_flags(codes, dp.n_pad);
                const size_t rb = dp.n_pad >> 3;
                v[k] = *reinterpret_cast<const uint4*>(codes + row + i0);
                sg[k] = fl[(size_t)w * rb + (i0 >> 3)];
                zr[k] = fl[(size_t)(DM17_W + w) * rb + (i0 >> 3)];
            }
        } else {
            v[k] = (i0 < dp.n_pad) ? *reinterpret_cast<const uint4*>(codes + row + i0)
                                   : make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
        }
    }
    // key e of load k: (valid, bucket inside the window, sign)
    auto decode = [&](int k, int e, uint32_t& bkt, uint32_t& neg) -> bool {
        const uint32_t wd[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        const uint32_t code = (wd[e >> 1] >> (16 * (e & 1))) & 0xffffu;
        if (WIDE) {
            bkt = code;
            neg = (sg[k] >> e) & 1u;
            return !((zr[k] >> e) & 1u);
        }
        neg = code >> 15;
        bkt = neg ? (code & 0x7fffu) - 1u : code;
        return code != DM_ZERO;
    };
    uint32_t pr[DM_PER1];   // partition << 16 | rank inside (tile, partition); 0xffffffff: no key
    const uint32_t submask = dp.SB - 1u;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int e = 0; e < G; ++e) {
            uint32_t bkt, neg;
            const uint32_t i = t0 + (k * DM_TB1 + tid) * G + e;
            const bool ok = decode(k, e, bkt, neg) && i < dp.n;
            const uint32_t p = bkt >> dp.sub_bits;
            pr[k * G + e] = ok ? ((p << 16) | atomicAdd(&cnt[p], 1u)) : 0xffffffffu;
        }
    }
    __syncthreads();
    for (uint32_t p = tid; p < dp.ppw; p += DM_TB1) {
        const uint32_t c = cnt[p];
        if (c) atomicAdd(&pcount[w * dp.ppw + p], c);
    }
    dm_block_scan(cnt, dp.ppw);   // cnt[p] = start of partition p, cnt[ppw] = keys in the tile
    uint32_t* tt = toff + ((size_t)w * dp.ntile + tile) * (dp.ppw + 1);
    for (uint32_t p = tid; p <= dp.ppw; p += DM_TB1) tt[p] = cnt[p];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int e = 0; e < G; ++e) {
            const uint32_t q = pr[k * G + e];
            if (q == 0xffffffffu) continue;
            uint32_t bkt, neg;
            decode(k, e, bkt, neg);
            const uint32_t i = t0 + (k * DM_TB1 + tid) * G + e;
            stage[cnt[q >> 16] + (q & 0xffffu)] = ((bkt & submask) << (dp.idx_bits + 1)) | (neg << dp.idx_bits) | i;
        }
    }
    __syncthreads();
    const uint32_t total = cnt[dp.ppw];
    uint32_t* out = items + (size_t)w * dp.n_row + t0;
    for (uint32_t k = 4 * tid; k < total; k += 4 * DM_TB1)   // rows and tiles start 16-byte aligned; the tail past `total` is slack
        *reinterpret_cast<uint4*>(out + k) = *reinterpret_cast<const uint4*>(stage + k);
}

// item -> bucket entry: point index | ENT_NEG | ENT_ENDO (GLV rows: positions >= n_pts are second halves)
FP_INLINE uint32_t dm_entry(uint32_t it, const DmPlan& dp, uint32_t idxmask) {
    uint32_t idx = it & idxmask;
    uint32_t e = ((it >> dp.idx_bits) & 1u) << 31;
    if (idx >= dp.n_pts) {
        idx -= dp.n_pts;
        e |= ENT_ENDO;
    }
    return e | idx;
}

// h[b] += 1 for every valid lane; returns the lane's slot (old value + its rank among the lanes counted with it).  The
// two most frequent-looking buckets of the wave (the first lane's, then the first remaining lane's) take one LDS atomic
// each, whatever is left goes lane by lane: skew collapses, spread buckets cost two wave-uniform rounds more.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
FP_INLINE uint32_t dm_wave_agg_add(lds_u32* h, uint32_t b, bool valid) {
    const uint32_t lane = __lane_id();
    uint64_t todo = __ballot(valid);
    uint32_t res = 0;
    bool pending = valid;
#pragma unroll 1
    for (int round = 0; round < 2 && todo; ++round) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t bl = __shfl(b, lead);
        const bool mine = pending && b == bl;
        const uint64_t same = __ballot(mine);
        uint32_t base = 0;
        if ((int)lane == lead)
            base = __hip_atomic_fetch_add(&h[bl], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __shfl(base, lead);
        if (mine) {
            res = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            pending = false;
        }
        todo &= ~same;
    }
    if (pending) res = __hip_atomic_fetch_add(&h[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return res;
}

// One pass over an over-long partition (skewed scalars: it does not fit the level-2 stage), tile-major: a wave per
// level-1 run, eight keys per lane in flight, one LDS atomic for a wave's dominant buckets (dm_wave_agg_add) — with all scalars equal a
// whole row lands in one bucket, where per-lane atomics on the one address serialise (and where the even-runs guess of
// the fast path's key -> tile lookup is off by half the row under GLV).  PLACE = false counts into h[], PLACE = true
// takes slots from h[] (bucket starts) and writes the entries.  Kept out of line: inlined, its registers cost the
// fast path its load batching (1.29 -> 1.55 ms per 2^20-point step, profiles/r02_sweeps.txt).
template <bool PLACE>
__device__ __attribute__((noinline)) void dm_long_pass(const uint32_t* __restrict__ rowp, const lds_u32* rstart,
                                                       const lds_u32* rpre, lds_u32* h, uint32_t* __restrict__ out,
                                                       uint32_t ntile, int idx_bits, uint32_t n_pts) {
    constexpr int LB = 8;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const int sh = idx_bits + 1;
    const uint32_t idxmask = (1u << idx_bits) - 1u;
    for (uint32_t t = wv; t < ntile; t += DM_TB2 / 64) {
        const uint32_t len = rpre[t + 1] - rpre[t];
        const uint32_t* src = rowp + (size_t)t * DM_T1 + rstart[t];
        for (uint32_t k0 = 0; k0 < len; k0 += 64 * LB) {
            uint32_t v[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const uint32_t k = k0 + j * 64 + lane;
                v[j] = k < len ? src[k] : 0u;
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const bool ok = k0 + j * 64 + lane < len;
                const uint32_t at = dm_wave_agg_add(h, v[j] >> sh, ok);
                if (PLACE && ok) {
                    uint32_t idx = v[j] & idxmask, e = ((v[j] >> idx_bits) & 1u) << 31;
                    if (idx >= n_pts) {
                        idx -= n_pts;
                        e |= ENT_ENDO;
                    }
                    out[at] = e | idx;
                }
            }
        }
    }
}

// level 2, one workgroup per partition (w, pl).  Key k of the partition (k < total) lives in tile t with
// rpre[t] <= k < rpre[t + 1]; runs are nearly equal, so t is guessed from k and corrected by a step or two.
// (waves_per_eu 6: PER = 16 then fits 66 VGPRs without scratch — three workgroups per CU alone, two beside the row / column sums)
template <int PER>
__global__ void __launch_bounds__(DM_TB2) __attribute__((amdgpu_waves_per_eu(6, 6))) k_dm_bucket_sort(const uint32_t* __restrict__ pstart,
                                                           const uint32_t* __restrict__ toff,
                                                           const uint32_t* __restrict__ items, DmPlan dp, uint32_t NB,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                           uint32_t* __restrict__ entries) {
    SORT_PRIO();
    constexpr int TB = DM_TB2;
    __shared__ uint32_t h[DM_MAX_PPW + 8];              // [SB + 1]
    __shared__ uint32_t rstart[DM_MAX_TILES];
    __shared__ uint32_t rpre[DM_MAX_TILES + 8];         // run lengths, then their exclusive scan
    constexpr int DM_STAGE = PER * DM_TB2;
    __shared__ __attribute__((aligned(16))) uint32_t stage[DM_STAGE];
    // last window first: its partitions are the long ones (14 bits: four times the keys), they should not be the kernel's tail
    const uint32_t p = gridDim.x - 1u - blockIdx.x, w = p / dp.ppw, pl = p - w * dp.ppw;
    const int tid = threadIdx.x;
    const uint32_t start = pstart[p], total = pstart[p + 1] - start;
    for (uint32_t b = tid; b <= dp.SB; b += TB) h[b] = 0;
    for (uint32_t t = tid; t < dp.ntile; t += TB) {
        const uint32_t* tt = toff + ((size_t)w * dp.ntile + t) * (dp.ppw + 1) + pl;
        const uint32_t a = tt[0];
        rstart[t] = a;
        rpre[t] = tt[1] - a;
    }
    dm_block_scan(rpre, dp.ntile);
    const uint32_t* rowp = items + (size_t)w * dp.n_row;
    const int sh = dp.idx_bits + 1;
    const uint32_t idxmask = (1u << dp.idx_bits) - 1u;
    const uint32_t key0 = w * NB + (pl << dp.sub_bits);
    const float guess = total ? (float)dp.ntile / (float)total : 0.f;
    auto locate = [&](uint32_t k) -> const uint32_t* {   // address of key k of this partition
        uint32_t t = (uint32_t)((float)k * guess);
        if (t >= dp.ntile) t = dp.ntile - 1;
        while (rpre[t] > k) --t;
        while (rpre[t + 1] <= k) ++t;
        return rowp + (size_t)t * DM_T1 + rstart[t] + (k - rpre[t]);
    };
    if (total <= (uint32_t)DM_STAGE) {
        uint32_t it[PER], rk[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t k = tid + j * TB;
            it[j] = k < total ? *locate(k) : 0u;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t k = tid + j * TB;
            rk[j] = k < total ? atomicAdd(&h[it[j] >> sh], 1u) : 0u;
        }
        __syncthreads();
        for (uint32_t b = tid; b < dp.SB; b += TB) hist[key0 + b] = h[b];
        dm_block_scan(h, dp.SB);
        for (uint32_t b = tid; b < dp.SB; b += TB) offs[key0 + b] = start + h[b];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t k = tid + j * TB;
            if (k < total) stage[h[it[j] >> sh] + rk[j]] = dm_entry(it[j], dp, idxmask);
        }
        __syncthreads();
        for (uint32_t k = tid; k < total; k += TB) entries[start + k] = stage[k];
        return;
    }
    if (total <= 8u * (uint32_t)DM_STAGE) {
        // over-long partition with buckets still spread (the top window's: its 14 bits give four times the keys per
        // bucket): count, scan, place straight into entries[]
        constexpr int MB = 4;   // keys per thread in flight
        for (uint32_t k0 = tid; k0 < total; k0 += MB * TB) {
            uint32_t v[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) v[j] = k0 + j * TB < total ? *locate(k0 + j * TB) : 0u;
#pragma unroll
            for (int j = 0; j < MB; ++j)
                if (k0 + j * TB < total) atomicAdd(&h[v[j] >> sh], 1u);
        }
        __syncthreads();
        for (uint32_t b = tid; b < dp.SB; b += TB) hist[key0 + b] = h[b];
        dm_block_scan(h, dp.SB);
        for (uint32_t b = tid; b < dp.SB; b += TB) offs[key0 + b] = start + h[b];
        __syncthreads();
        for (uint32_t k0 = tid; k0 < total; k0 += MB * TB) {
            uint32_t v[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) v[j] = k0 + j * TB < total ? *locate(k0 + j * TB) : 0u;
#pragma unroll
            for (int j = 0; j < MB; ++j)
                if (k0 + j * TB < total) entries[start + atomicAdd(&h[v[j] >> sh], 1u)] = dm_entry(v[j], dp, idxmask);
        }
        return;
    }
    // more than eight stages: skewed scalars (dm_long_pass)
    dm_long_pass<false>(rowp, (const lds_u32*)rstart, (const lds_u32*)rpre, (lds_u32*)h, nullptr, dp.ntile, dp.idx_bits, dp.n_pts);
    __syncthreads();
    for (uint32_t b = tid; b < dp.SB; b += TB) hist[key0 + b] = h[b];
    dm_block_scan(h, dp.SB);
    for (uint32_t b = tid; b < dp.SB; b += TB) offs[key0 + b] = start + h[b];
    __syncthreads();
    dm_long_pass<true>(rowp, (const lds_u32*)rstart, (const lds_u32*)rpre, (lds_u32*)h, entries + start, dp.ntile, dp.idx_bits, dp.n_pts);
}

// ------------------------------------------------------------------ order buckets by length (descending)
FP_INLINE uint32_t size_bin(uint32_t len) { return (SIZE_BINS - 1) - (len < SIZE_BINS ? len : SIZE_BINS - 1); }

__global__ void __launch_bounds__(BLOCK) k_size_count(const uint32_t* __restrict__ hist, uint32_t nbt,
                                                      uint32_t* __restrict__ bin_count) {
    SORT_PRIO();
    __shared__ uint32_t cnt[SIZE_BINS];
    for (uint32_t b = threadIdx.x; b < SIZE_BINS; b += BLOCK) cnt[b] = 0;
    __syncthreads();
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < nbt; k += gridDim.x * BLOCK)
        atomicAdd(&cnt[size_bin(hist[k])], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < SIZE_BINS; b += BLOCK)
        if (cnt[b]) atomicAdd(&bin_count[b], cnt[b]);
}
__global__ void __launch_bounds__(BLOCK) k_size_scatter(const uint32_t* __restrict__ hist, uint32_t nbt,
                                                        uint32_t* __restrict__ bin_cursor, uint32_t* __restrict__ order) {
    SORT_PRIO();
    __shared__ uint32_t cnt[SIZE_BINS];
    __shared__ uint32_t baseb[SIZE_BINS];
    for (uint32_t b = threadIdx.x; b < SIZE_BINS; b += BLOCK) cnt[b] = 0;
    __syncthreads();
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < nbt; k += gridDim.x * BLOCK)
        atomicAdd(&cnt[size_bin(hist[k])], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < SIZE_BINS; b += BLOCK) {
        const uint32_t v = cnt[b];
        baseb[b] = v ? atomicAdd(&bin_cursor[b], v) : 0u;
        cnt[b] = 0;
    }
    __syncthreads();
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < nbt; k += gridDim.x * BLOCK) {
        const uint32_t b = size_bin(hist[k]);
        order[baseb[b] + atomicAdd(&cnt[b], 1u)] = k;
    }
}

}  // namespace h2agg
