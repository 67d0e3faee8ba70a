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
#include "batch_kernels.cuh"

namespace h2agg {

constexpr int SORT_MAX_PW = 1024;   // partitions (LDS counters in level 1)
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
};

// Signed-digit recode of a canonical scalar; calls f(window, bucket_index(0-based), negative) for every
// non-zero digit.  Digits lie in [-2^(c-1), 2^(c-1)]; W*c >= 255 guarantees no carry out of the top window.
template <class F>
FP_INLINE void msm_for_each_digit(U256 s, int c, int W, F&& f) {
    const uint32_t mask = (1u << c) - 1u;
    const uint32_t half = 1u << (c - 1);
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
        if (mag != 0) f(w, mag - 1u, neg);
    }
}

// ------------------------------------------------------------------ level 1
__global__ void __launch_bounds__(BLOCK) k_part_count(const uint8_t* __restrict__ scalars, size_t n, int c, int W,
                                                      SortPlan sp, uint32_t* __restrict__ pcount, uint32_t* flags) {
    __shared__ uint32_t cnt[SORT_MAX_PW];
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) cnt[p] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * sp.tile;
    uint32_t bad = 0;
    for (uint32_t k = threadIdx.x; k < sp.tile; k += BLOCK) {
        const size_t i = base + k;
        if (i >= n) break;
        U256 s = u256_load(scalars + 32 * i);
        bad |= !u256_is_canonical_fr(s);
        msm_for_each_digit(s, c, W, [&](int w, uint32_t b, bool) {
            atomicAdd(&cnt[(uint32_t)w * sp.ppw + (b >> sp.sub_bits)], 1u);
        });
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) {
        const uint32_t v = cnt[p];
        if (v) atomicAdd(&pcount[p], v);
    }
}

// exclusive scan of up to 2 * SORT_MAX_PW values by one workgroup: out[i] = sum_{j<i} in[j], out[n] = total;
// cursor (optional) receives a copy of out[0..n)
__global__ void __launch_bounds__(BLOCK) k_scan_small(const uint32_t* __restrict__ in, uint32_t n,
                                                      uint32_t* __restrict__ out, uint32_t* __restrict__ cursor) {
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
    for (uint32_t k = threadIdx.x; k < sp.tile; k += BLOCK) {
        const size_t i = base + k;
        if (i >= n) break;
        U256 s = u256_load(scalars + 32 * i);
        msm_for_each_digit(s, c, W, [&](int w, uint32_t b, bool) {
            atomicAdd(&cnt[(uint32_t)w * sp.ppw + (b >> sp.sub_bits)], 1u);
        });
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < sp.PW; p += BLOCK) {
        const uint32_t v = cnt[p];
        basep[p] = v ? atomicAdd(&pcursor[p], v) : 0u;  // one device-scope atomic per (workgroup, partition)
        cnt[p] = 0;
    }
    __syncthreads();
    const uint32_t submask = sp.SB - 1u;
    for (uint32_t k = threadIdx.x; k < sp.tile; k += BLOCK) {
        const size_t i = base + k;
        if (i >= n) break;
        U256 s = u256_load(scalars + 32 * i);
        msm_for_each_digit(s, c, W, [&](int w, uint32_t b, bool neg) {
            const uint32_t p = (uint32_t)w * sp.ppw + (b >> sp.sub_bits);
            const uint32_t pos = basep[p] + atomicAdd(&cnt[p], 1u);
            item_idx[pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
            item_sub[pos] = (uint16_t)(b & submask);
        });
    }
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
    for (uint32_t k = start + tid; k < end; k += BLOCK) atomicAdd(&h[item_sub[k]], 1u);
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
    for (uint32_t k = start + tid; k < end; k += BLOCK) {
        const uint32_t r = atomicAdd(&h[item_sub[k]], 1u);
        entries[start + r] = item_idx[k];
    }
}

// ------------------------------------------------------------------ order buckets by length (descending)
FP_INLINE uint32_t size_bin(uint32_t len) { return (SIZE_BINS - 1) - (len < SIZE_BINS ? len : SIZE_BINS - 1); }

__global__ void __launch_bounds__(BLOCK) k_size_count(const uint32_t* __restrict__ hist, uint32_t nbt,
                                                      uint32_t* __restrict__ bin_count) {
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
