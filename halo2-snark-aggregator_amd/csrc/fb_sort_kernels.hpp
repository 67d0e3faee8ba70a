// Sort for MSMs over a table with fixed-base levels at c = 20 (h2agg_bases_precompute on 2^18 .. 2^22-point tables: the
// g_lagrange of BASELINE.json configs[4], assign_instance_commitment verify.rs:601-603,623-640).
//
// With levels 2^(20 w) P_i stored for w < 13, ALL 13 signed 20-bit digits of a scalar land in ONE set of 2^19 buckets: 13
// insertions per scalar instead of the 15 of the ordinary path at c = 17, one bucket reduction per MSM, no doubling chain.  A
// bucket entry then names (level, point): 26 bits at 2^22 points, which the packed item of sort_kernels.hpp (sub-bucket | sign |
// index in 32 bits) cannot carry next to 9 sub-bucket bits — round 4 measured the two-array fallback at 2.1 ms of sort per
// 2^22-point MSM and refused such tables.  Here the item only travels INSIDE a tile, so it names its scalar tile-locally:
//
//   level 1   k_fb_partition: a workgroup takes a tile of 2048 SCALARS (not keys of one window: all levels share the bucket
//             space, so nothing separates them), recodes each into 13 digits in registers and orders the tile's 26 624 keys by
//             partition in 104 KiB of LDS: one returning LDS atomic per key; the tile goes back coalesced with its table of
//             2305 partition offsets.  item = low slot bits : 8 (10) | sign | level : 4 | scalar : 11          (0.12 ms at 2^22)
//   level 2   k_fb_bucket_sort: one workgroup per partition (256 buckets) collects its ~12-key run from every tile and orders
//             it by (bucket, level) in LDS — 3 328 counters, one atomic per key — then writes entries[] / hist[] / offs[] of its
//             buckets.  entry = sign << 31 | level * n_level + point.                                          (0.39-0.42 ms)
//
// What bounds level 2 (profiles/r05_sweeps.txt section 1): 2 048 runs of ~50 bytes per workgroup, each its own memory
// request; with every key's load in flight at once (LDS-DMA, below) a workgroup still waits ~14 us for them — the CU's
// outstanding-request budget at HBM latency — and a 154-KiB workgroup has the CU to itself, so nothing hides it.
// Entries of a bucket come out in level order; that was meant to keep the accumulation's gathers inside one or two levels at
// a time, and measured as no gain (3.95 ms either way: every (level, point) is read exactly once per MSM, there is nothing
// for a cache to keep) — it is simply the order the counters give.
#pragma once
#include "sort_kernels.hpp"

namespace h2agg {

constexpr int FB_C = 20;                       // window bits of the big-table levels
constexpr int FB_W = 13;                       // ceil(255 / 20) digit positions = levels
constexpr uint32_t FB_NB = 1u << (FB_C - 1);   // buckets of the one bucket set
constexpr int FB_T = 2048;                     // scalars per level-1 tile
constexpr int FB_TB1 = 1024;                   // level-1 threads (two scalars each)
constexpr int FB_SUB_BITS = 8;                 // low bucket bits resolved in level 2
constexpr uint32_t FB_SB = 1u << FB_SUB_BITS;
constexpr uint32_t FB_PPW = FB_NB >> FB_SUB_BITS;   // 2048 partitions
// The top digit has 14 bits (254 = 12 * 20 + 14): dropped into the same buckets it would put n / 2^14 extra entries into each
// of the lowest 2^14 — 3.7 x the mean run, on lanes that then outlast the launch (single 2^22-point MSM: accumulation 4.69 ms;
// with the slots below 3.95).  It gets bucket slots of its own instead: value m of the top digit, scalar i -> slot
// FB_NB + fb_xslot(m - 1, i & 15); k_fb_fold (msm_kernels.hpp) adds the sixteen parts of value m into bucket m - 1 in front of
// the reduction (same weight).  Sixteen parts make 3 x 2^18 slots = three full rounds of the chip's 4 096 wave slots; four
// parts (2.25 rounds) measured the same within a box's noise — longest-first order already puts the shortest runs last.
constexpr int FB_XPARTS_LOG = 4;
constexpr uint32_t FB_XB = (1u << 14) << FB_XPARTS_LOG;          // extra slots (2^18)
constexpr uint32_t FB_NBT = FB_NB + FB_XB;                        // bucket slots the accumulation walks
// partitions of the extra slots are four times as wide (1024 slots, one level: 1024 counters in level 2 against 256 x 13) so
// that their runs in a level-1 tile stay at 8 keys
constexpr int FB_XSUB_BITS = 10;
constexpr uint32_t FB_XSB = 1u << FB_XSUB_BITS;
constexpr uint32_t FB_NPART = FB_PPW + (FB_XB >> FB_XSUB_BITS);   // 2048 + 256 partitions
constexpr int FB_KEYS1 = FB_T * FB_W;          // keys of a tile (26 624)
constexpr int FB_MAX_TILES = 2048;             // n <= 2^22
constexpr int FB_TB2 = 1024;                   // level-2 threads
constexpr int FB_PER2 = 28;                    // keys per thread staged in LDS by level 2
constexpr int FB_STAGE = FB_PER2 * FB_TB2;     // 28 672 keys (112 KiB): a partition holds 12 n / 2048 <= 24 576 on average
constexpr uint32_t FB_KEYS2 = FB_SB * FB_W;    // level-2 counters: (bucket, level)

// exclusive scan of v[0 .. cnt) in LDS (cnt <= 4 * blockDim.x, blockDim.x = 1024), total -> v[cnt]; ws: 17 words of LDS
FP_INLINE void fb_block_scan(uint32_t* v, uint32_t cnt, uint32_t* ws) {
    __syncthreads();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t x[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x[j] = (4 * tid + j < cnt) ? v[4 * tid + j] : 0u;
        sum += x[j];
    }
    const uint32_t inc = dm_wave_scan_incl(sum, lane);
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wave; ++k) base += ws[k];
    uint32_t off = base + inc - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * tid + j <= cnt) v[4 * tid + j] = off;
        off += x[j];
    }
    if (4 * tid + 4 == cnt) v[cnt] = off;   // cnt = 4 * blockDim.x: the total has no thread of its own
    __syncthreads();
}

// the 13 signed 20-bit digits of a canonical scalar: f(w, bucket, neg) for every non-zero one.  r < 2^254: the top digit
// (bits 240 .. 253 + carry) never exceeds 2^19, so nothing is carried out.
template <class F>
FP_INLINE void fb_for_each_digit(const U256& s, F&& f) {
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < FB_W; ++w) {
        const int pos = FB_C * w, wd = pos >> 5, sh = pos & 31;
        uint32_t v = s.w[wd] >> sh;
        if (sh > 32 - FB_C && wd + 1 < 8) v |= s.w[wd + 1] << (32 - sh);
        const uint32_t raw = (v & ((1u << FB_C) - 1u)) + carry;
        const bool neg = raw > (1u << (FB_C - 1));
        carry = neg ? 1u : 0u;
        const uint32_t mag = neg ? (1u << FB_C) - raw : raw;
        f(w, mag - 1u, neg, mag != 0);
    }
}
// digit w of the tile's scalar `local`: its partition, and the low bits of its bucket slot inside the partition (the top
// digit's own slots: above)
// The top digit's slots are dealt to their 256 partitions by the LOW bits of the digit's value (slot of value m, part j:
// fb_xslot below): whatever range the top digit covers — 12 388 values for scalars uniform below r, 2^13 for "random bytes with
// the top three bits cleared" (bench.py's instance scalars), a few dozen for short scalars — its keys spread over all 256
// partitions.  Dealt by the HIGH bits (round 5) the 2^13-value case filled half the partitions with 32 k keys each, more than the
// level-2 stage holds: every one of them took the over-long path (0.46 instead of 0.39 ms per 2^22-point MSM).
constexpr uint32_t FB_XPART_COUNT = FB_XB >> FB_XSUB_BITS;   // 256
FP_INLINE uint32_t fb_xslot(uint32_t bkt, uint32_t part) {   // slot (beyond FB_NB) of value bkt + 1, part `part`
    return (bkt & (FB_XPART_COUNT - 1u)) * FB_XSB + (((bkt / FB_XPART_COUNT) << FB_XPARTS_LOG) | part);
}
FP_INLINE uint32_t fb_part(int w, uint32_t bkt) {
    return w == FB_W - 1 ? FB_PPW + (bkt & (FB_XPART_COUNT - 1u)) : bkt >> FB_SUB_BITS;
}
FP_INLINE uint32_t fb_sub(int w, uint32_t bkt, uint32_t local) {
    return w == FB_W - 1 ? fb_xslot(bkt, local & ((1u << FB_XPARTS_LOG) - 1u)) & (FB_XSB - 1u) : bkt & (FB_SB - 1u);
}

// grid: ceil(n / FB_T) tiles.  items[tile * FB_KEYS1 + k], k < toff[tile][FB_PPW]: the tile's keys ordered by partition;
// toff[tile * (FB_NPART + 1) + p]: where partition p starts inside the tile; pcount[p] += its length
__global__ void __launch_bounds__(FB_TB1) k_fb_partition(const uint8_t* __restrict__ scalars, uint32_t n,
                                                         uint32_t* __restrict__ pcount, uint32_t* __restrict__ toff,
                                                         uint32_t* __restrict__ items, uint32_t* flags) {
    SORT_PRIO();
    __shared__ uint32_t cnt[FB_NPART + 1];
    __shared__ uint32_t ws[17];
    __shared__ __attribute__((aligned(16))) uint32_t stage[FB_KEYS1];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    for (uint32_t p = tid; p <= FB_NPART; p += FB_TB1) cnt[p] = 0;
    __syncthreads();
    constexpr int SPT = FB_T / FB_TB1;   // scalars per thread
    U256 s[SPT];
    bool live[SPT];
    uint32_t bad = 0;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const uint32_t i = tile * FB_T + j * FB_TB1 + tid;
        live[j] = i < n;
        if (live[j]) s[j] = u256_load(scalars + 32 * (size_t)i);
    }
    uint32_t pr[SPT * FB_W];   // partition << 16 | rank inside (tile, partition); 0xffffffff: no key
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        // a scalar >= r is refused by the call (FLAG_NONCANONICAL) and must not reach the digit loops: with bits 254 / 255
        // set its top digit exceeds 2^14 and names a partition past FB_NPART — LDS counters out of bounds
        if (live[j] && !u256_is_canonical_fr(s[j])) {
            bad = 1;
            live[j] = false;
        }
        if (!live[j]) {
#pragma unroll
            for (int w = 0; w < FB_W; ++w) pr[j * FB_W + w] = 0xffffffffu;
            continue;
        }
        fb_for_each_digit(s[j], [&](int w, uint32_t bkt, bool, bool ok) {
            const uint32_t p = fb_part(w, bkt);
            pr[j * FB_W + w] = ok ? ((p << 16) | atomicAdd(&cnt[p], 1u)) : 0xffffffffu;
        });
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    __syncthreads();
    for (uint32_t p = tid; p < FB_NPART; p += FB_TB1) {
        const uint32_t c = cnt[p];
        if (c) atomicAdd(&pcount[p], c);
    }
    fb_block_scan(cnt, FB_NPART, ws);   // cnt[p] = start of partition p, cnt[FB_NPART] = keys in the tile
    uint32_t* tt = toff + (size_t)tile * (FB_NPART + 1);
    for (uint32_t p = tid; p <= FB_NPART; p += FB_TB1) tt[p] = cnt[p];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        if (!live[j]) continue;
        fb_for_each_digit(s[j], [&](int w, uint32_t bkt, bool neg, bool) {
            const uint32_t q = pr[j * FB_W + w];
            if (q != 0xffffffffu)
                stage[cnt[q >> 16] + (q & 0xffffu)] = (fb_sub(w, bkt, (uint32_t)(j * FB_TB1) + tid) << 16) | ((neg ? 1u : 0u) << 15) |
                                                      ((uint32_t)w << 11) | (uint32_t)(j * FB_TB1 + tid);
        });
    }
    __syncthreads();
    const uint32_t total = cnt[FB_NPART];
    uint32_t* out = items + (size_t)tile * FB_KEYS1;
    for (uint32_t k = 4 * tid; k < total; k += 4 * FB_TB1)   // tiles start 16-byte aligned; the tail past `total` is slack inside the tile
        *reinterpret_cast<uint4*>(out + k) = *reinterpret_cast<const uint4*>(stage + k);
}

// One pass over an over-long partition, tile-major (see k_fb_bucket_sort).  PLACE = false counts into h[] ((bucket, level)
// counters, or the extra slots' own); PLACE = true takes positions from h[] (the counters' scan) and writes the entries.  Kept
// out of line: the fast path's registers are spoken for.
template <bool PLACE>
FP_INLINE void fb_long_pass(const uint32_t* __restrict__ items, const lds_u32* rstart, const lds_u32* rpre,
                                                       lds_u32* h, uint32_t* __restrict__ out, uint32_t ntile, uint32_t n_level, bool ext) {
    constexpr int LB = 8;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t t = wv; t < ntile; t += FB_TB2 / 64) {
        const uint32_t len = rpre[t + 1] - rpre[t];
        const uint32_t* src = items + (size_t)t * FB_KEYS1 + rstart[t];
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
                const uint32_t it = v[j];
                const uint32_t key = ext ? it >> 16 : (it >> 16) * FB_W + ((it >> 11) & 15u);
                const uint32_t at = dm_wave_agg_add(h, key, ok);
                if (PLACE && ok) out[at] = ((it >> 15) & 1u) << 31 | (((it >> 11) & 15u) * n_level + t * FB_T + (it & 2047u));
            }
        }
    }
}

// level 2, one workgroup per partition p.  Key k of the partition (k < total) lives in tile t with rpre[t] <= k < rpre[t + 1].
__global__ void __launch_bounds__(FB_TB2) k_fb_bucket_sort(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ toff,
                                                           const uint32_t* __restrict__ items, uint32_t n_level, uint32_t ntile,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                           uint32_t* __restrict__ entries) {
    SORT_PRIO();
    constexpr int TB = FB_TB2;
    __shared__ uint32_t h[FB_KEYS2 + 8];
    __shared__ uint32_t ws[17];
    __shared__ uint32_t rstart[FB_MAX_TILES];
    __shared__ uint32_t rpre[FB_MAX_TILES + 8];
    __shared__ __attribute__((aligned(16))) uint32_t stage[FB_STAGE];
    __shared__ uint32_t longt[FB_STAGE / 64 + 1];
    __shared__ uint32_t nlong;
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t start = pstart[p], total = pstart[p + 1] - start;
    for (uint32_t b = tid; b <= FB_KEYS2; b += TB) h[b] = 0;
    if (tid == 0) nlong = 0;
    for (uint32_t t = tid; t < ntile; t += TB) {
        const uint32_t* tt = toff + (size_t)t * (FB_NPART + 1) + p;
        const uint32_t a = tt[0];
        rstart[t] = a;
        rpre[t] = tt[1] - a;
    }
    fb_block_scan(rpre, ntile, ws);
    // an ordinary partition: 256 buckets x 13 levels; a partition of the top digit's slots: 1024 slots, that one level
    const bool ext = p >= FB_PPW;
    const uint32_t key0 = ext ? FB_NB + ((p - FB_PPW) << FB_XSUB_BITS) : p << FB_SUB_BITS;
    const uint32_t nslot = ext ? FB_XSB : FB_SB, lv = ext ? 1u : (uint32_t)FB_W, nkeys = nslot * lv;
    // item (low slot bits : 8 or 10 | sign | level : 4 | scalar : 11) of tile t -> entry, counter index
    auto entry_of = [&](uint32_t it, uint32_t t) -> uint32_t {
        return ((it >> 15) & 1u) << 31 | (((it >> 11) & 15u) * n_level + t * FB_T + (it & 2047u));
    };
    auto key_of = [&](uint32_t it) -> uint32_t { return ext ? it >> 16 : (it >> 16) * FB_W + ((it >> 11) & 15u); };
    auto write_hist = [&]() {   // hist / offs of the partition's buckets from the (bucket, level) counters; h[] becomes its scan
        __syncthreads();
        for (uint32_t b = tid; b < nslot; b += TB) {
            uint32_t sum = 0;
            for (uint32_t w = 0; w < lv; ++w) sum += h[b * lv + w];
            hist[key0 + b] = sum;
        }
        fb_block_scan(h, nkeys, ws);
        for (uint32_t b = tid; b < nslot; b += TB) offs[key0 + b] = start + h[b * lv];
    };
    if (total == 0) {   // (a partition nobody has a key for: tiny MSMs)
        write_hist();
        return;
    }
    if (total > (uint32_t)FB_STAGE) return;
    {
        // Where key k lives: a search over rpre[] per key (eleven dependent LDS reads) was two thirds of the kernel.  The runs'
        // owners say it instead: the thread that loaded tile t's offsets writes the source index of every key of its run into
        // the key's slot of the stage (a dozen words); each key's thread reads its slot and replaces it with the key itself by
        // LDS-DMA (one dword per lane from its own address, a wave's 64 as one 256-byte row, no registers, thirty in flight per
        // lane).  Runs longer than 64 keys (skewed scalars) are finished by whole waves.
        for (uint32_t t = tid; t < ntile; t += TB) {
            const uint32_t k0 = rpre[t], len = rpre[t + 1] - k0, src0 = t * (uint32_t)FB_KEYS1 + rstart[t];
            const uint32_t own = len < 64u ? len : 64u;
            for (uint32_t i = 0; i < own; ++i) stage[k0 + i] = src0 + i;
            if (len > 64u) longt[atomicAdd(&nlong, 1u)] = t;
        }
        __syncthreads();
        for (uint32_t q = tid >> 6; q < nlong; q += TB / 64) {
            const uint32_t t = longt[q], k0 = rpre[t], len = rpre[t + 1] - k0, src0 = t * (uint32_t)FB_KEYS1 + rstart[t];
            for (uint32_t i = 64u + (tid & 63u); i < len; i += 64u) stage[k0 + i] = src0 + i;
        }
        if (nlong) __syncthreads();
        uint32_t ent[FB_PER2], kr[FB_PER2];
        const uint32_t wave_base = tid & ~63u;
#pragma unroll
        for (int j = 0; j < FB_PER2; ++j) {   // (row j of a wave is read and then overwritten by that wave alone: program order)
            const uint32_t k = tid + j * TB;
            kr[j] = k < total ? stage[k] : 0u;   // (past the end: any valid address; the slot of another row may already hold a key)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(items + kr[j]),
                                             (__attribute__((address_space(3))) void*)(stage + wave_base + j * TB), 4, 0, 0);
            asm volatile("" ::: "memory");
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FB_PER2; ++j) {
            const uint32_t it = stage[tid + j * TB];
            ent[j] = entry_of(it, kr[j] / (uint32_t)FB_KEYS1);
            kr[j] = key_of(it) << 16;
            if (j % 6 == 5) asm volatile("" ::: "memory");   // (six LDS reads in flight, not thirty: the register budget)
        }
#pragma unroll
        for (int j = 0; j < FB_PER2; ++j) {
            const uint32_t k = tid + j * TB;
            if (k < total) kr[j] |= atomicAdd(&h[kr[j] >> 16], 1u);
            if (j % 6 == 5) asm volatile("" ::: "memory");
        }
        write_hist();
#pragma unroll
        for (int j = 0; j < FB_PER2; ++j) {
            const uint32_t k = tid + j * TB;
            if (k < total) stage[h[kr[j] >> 16] + (kr[j] & 0xffffu)] = ent[j];
        }
        __syncthreads();
        for (uint32_t k = tid; k < total; k += TB) entries[start + k] = stage[k];
        return;
    }
    // (over-long partitions: k_fb_bucket_sort_long, launched behind this kernel)
}

// ---- very long partitions across workgroups.  One digit of SMALL scalars (64-bit instance values: digit 3 has four bits) puts
// every key of its level into the first sixteen buckets — 2^22 keys in partition 0, which one workgroup orders in 4.3 ms (LDS
// atomics of one CU) while 255 CUs wait: 16 MSMs of 64-bit scalars over a 2^22-point table 93 ms with levels against 50 on the
// ordinary path.  Partitions whose runs are long (the tile-major shape, >= 48 keys per tile on average; at most FB_LONG_CAP of
// them, listed by k_fb_scan_list) are therefore split over FB_LONG_S workgroups by tile range:
//   k_fb_long_count   (idx, y): counts its tiles' keys per (bucket, level) in LDS, stores the counters; the LAST of a
//                     partition's workgroups to finish (a ticket) sums them, writes hist / offs and turns every workgroup's
//                     counters into its starting positions
//   k_fb_long_place   (idx, y): takes positions from its counters (LDS atomics) and writes the entries
// Workgroups beyond the list's length leave at once: three short launches on inputs that need none of this.
constexpr uint32_t FB_LONG_CAP = 64, FB_LONG_S = 16;
constexpr uint32_t FB_LONG_NONE = 0xffffffffu;
// layout of the scratch block (words)
constexpr size_t FB_LONG_TICKET = 1, FB_LONG_LIST = FB_LONG_TICKET + FB_LONG_CAP, FB_LONG_IDX = FB_LONG_LIST + FB_LONG_CAP,
                 FB_LONG_COUNTS = (FB_LONG_IDX + FB_NPART + 63) & ~(size_t)63,
                 FB_LONG_WORDS = FB_LONG_COUNTS + (size_t)FB_LONG_CAP * FB_LONG_S * FB_KEYS2;
FP_INLINE bool fb_is_long(uint32_t total, uint32_t ntile) { return total > (uint32_t)FB_STAGE && total / ntile >= 48u; }

// pstart[] = exclusive scan of pcount[0 .. FB_NPART) (as k_dm_scan), and the list of very long partitions
__global__ void __launch_bounds__(1024) k_fb_scan_list(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t ntile,
                                                       uint32_t* __restrict__ lng) {
    SORT_PRIO();
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t nl;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = FB_NPART;
    if (tid == 0) nl = 0;
    if (tid < FB_LONG_CAP) lng[FB_LONG_TICKET + tid] = 0;
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
        const uint32_t p = 8 * tid + j;
        if (p <= n) out[p] = off;
        off += x[j];
        if (p < n) {
            uint32_t idx = FB_LONG_NONE;
            if (fb_is_long(x[j], ntile)) {
                idx = atomicAdd(&nl, 1u);
                if (idx < FB_LONG_CAP) lng[FB_LONG_LIST + idx] = p;
                else idx = FB_LONG_NONE;   // (beyond the list: the one-workgroup kernel takes it)
            }
            lng[FB_LONG_IDX + p] = idx;
        }
    }
    __syncthreads();
    if (tid == 0) lng[0] = nl < FB_LONG_CAP ? nl : FB_LONG_CAP;
}

// tiles [t0, t1) of partition p, tile-major (as fb_long_pass, the runs read off toff[] directly)
template <bool PLACE>
FP_INLINE void fb_long_tiles(const uint32_t* __restrict__ items, const uint32_t* __restrict__ toff, uint32_t p, uint32_t t0, uint32_t t1,
                             lds_u32* h, uint32_t* __restrict__ out, uint32_t n_level, bool ext) {
    constexpr int LB = 8;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t t = t0 + wv; t < t1; t += FB_TB2 / 64) {
        const uint32_t* tt = toff + (size_t)t * (FB_NPART + 1) + p;
        const uint32_t a = tt[0], len = tt[1] - a;
        const uint32_t* src = items + (size_t)t * FB_KEYS1 + a;
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
                const uint32_t it = v[j];
                const uint32_t key = ext ? it >> 16 : (it >> 16) * FB_W + ((it >> 11) & 15u);
                const uint32_t at = dm_wave_agg_add(h, key, ok);
                if (PLACE && ok) out[at] = ((it >> 15) & 1u) << 31 | (((it >> 11) & 15u) * n_level + t * FB_T + (it & 2047u));
            }
        }
    }
}

__global__ void __launch_bounds__(FB_TB2) k_fb_long_count(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ toff,
                                                          const uint32_t* __restrict__ items, uint32_t n_level, uint32_t ntile,
                                                          uint32_t* __restrict__ lng, uint32_t* __restrict__ hist, uint32_t* __restrict__ offs) {
    constexpr int TB = FB_TB2;
    const uint32_t idx = blockIdx.x, y = blockIdx.y, tid = threadIdx.x;
    if (idx >= lng[0]) return;
    SORT_PRIO();
    __shared__ uint32_t h[FB_KEYS2 + 8];
    __shared__ uint32_t ws[17];
    __shared__ uint32_t last;
    const uint32_t p = lng[FB_LONG_LIST + idx];
    const bool ext = p >= FB_PPW;
    const uint32_t key0 = ext ? FB_NB + ((p - FB_PPW) << FB_XSUB_BITS) : p << FB_SUB_BITS;
    const uint32_t nslot = ext ? FB_XSB : FB_SB, lv = ext ? 1u : (uint32_t)FB_W, nkeys = nslot * lv;
    for (uint32_t b = tid; b <= FB_KEYS2; b += TB) h[b] = 0;
    __syncthreads();
    const uint32_t t0 = (uint32_t)((uint64_t)ntile * y / FB_LONG_S), t1 = (uint32_t)((uint64_t)ntile * (y + 1) / FB_LONG_S);
    fb_long_tiles<false>(items, toff, p, t0, t1, (lds_u32*)h, nullptr, n_level, ext);
    __syncthreads();
    uint32_t* mine = lng + FB_LONG_COUNTS + ((size_t)idx * FB_LONG_S + y) * FB_KEYS2;
    for (uint32_t k = tid; k < nkeys; k += TB) mine[k] = h[k];
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(&lng[FB_LONG_TICKET + idx], 1u) == FB_LONG_S - 1u ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    // the partition's last workgroup: totals per counter, hist / offs, and every workgroup's starting positions
    __threadfence();
    uint32_t* all = lng + FB_LONG_COUNTS + (size_t)idx * FB_LONG_S * FB_KEYS2;
    for (uint32_t k = tid; k < nkeys; k += TB) {
        uint32_t sum = 0;
        for (uint32_t yy = 0; yy < FB_LONG_S; ++yy)
            sum += __hip_atomic_load(all + (size_t)yy * FB_KEYS2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h[k] = sum;
    }
    for (uint32_t k = nkeys + tid; k <= FB_KEYS2; k += TB) h[k] = 0;
    __syncthreads();
    const uint32_t start = pstart[p];
    for (uint32_t b = tid; b < nslot; b += TB) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < lv; ++w) sum += h[b * lv + w];
        hist[key0 + b] = sum;
    }
    fb_block_scan(h, nkeys, ws);
    for (uint32_t b = tid; b < nslot; b += TB) offs[key0 + b] = start + h[b * lv];
    for (uint32_t k = tid; k < nkeys; k += TB) {
        uint32_t at = h[k];
        for (uint32_t yy = 0; yy < FB_LONG_S; ++yy) {
            uint32_t* q = all + (size_t)yy * FB_KEYS2 + k;
            const uint32_t cnt = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *q = at;
            at += cnt;
        }
    }
}

__global__ void __launch_bounds__(FB_TB2) k_fb_long_place(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ toff,
                                                          const uint32_t* __restrict__ items, uint32_t n_level, uint32_t ntile,
                                                          const uint32_t* __restrict__ lng, uint32_t* __restrict__ entries) {
    constexpr int TB = FB_TB2;
    const uint32_t idx = blockIdx.x, y = blockIdx.y, tid = threadIdx.x;
    if (idx >= lng[0]) return;
    SORT_PRIO();
    __shared__ uint32_t h[FB_KEYS2 + 8];
    const uint32_t p = lng[FB_LONG_LIST + idx];
    const bool ext = p >= FB_PPW;
    const uint32_t nkeys = ext ? FB_XSB : FB_SB * (uint32_t)FB_W;
    const uint32_t* mine = lng + FB_LONG_COUNTS + ((size_t)idx * FB_LONG_S + y) * FB_KEYS2;
    for (uint32_t k = tid; k < nkeys; k += TB) h[k] = mine[k];
    __syncthreads();
    const uint32_t t0 = (uint32_t)((uint64_t)ntile * y / FB_LONG_S), t1 = (uint32_t)((uint64_t)ntile * (y + 1) / FB_LONG_S);
    fb_long_tiles<true>(items, toff, p, t0, t1, (lds_u32*)h, entries + pstart[p], n_level, ext);
}

// Over-long partitions (more keys than the level-2 stage holds: skewed scalars — small ones above all, where every key of a
// scalar's top non-zero digit lands in the first few buckets, i.e. in partition 0): same grid as k_fb_bucket_sort, launched
// behind it; a workgroup whose partition fitted the stage leaves at once.  Count, scan, place straight into entries[],
// TILE-major — a wave per level-1 run, read contiguously, eight keys per lane in flight; a wave's dominant counters take one
// LDS atomic (dm_wave_agg_add: with all scalars equal a partition's keys sit on 13 counters).  A kernel of its own: as a
// noinline call inside k_fb_bucket_sort it cost the fast path a third of its speed (0.46 -> 0.75 ms per 2^22-point MSM; the
// call's register convention), and round 5's inline version walked the keys in partition order with an eleven-step search per
// key — 10 ms for the 2^22 keys small scalars put into partition 0: 16 MSMs of 64-bit scalars over a 2^22-point table 202 ms
// against the ordinary path's 50 (profiles/r06_sweeps.txt section 4).
__global__ void __launch_bounds__(FB_TB2) k_fb_bucket_sort_long(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ toff,
                                                                const uint32_t* __restrict__ items, uint32_t n_level, uint32_t ntile,
                                                                const uint32_t* __restrict__ lng, uint32_t* __restrict__ hist,
                                                                uint32_t* __restrict__ offs, uint32_t* __restrict__ entries) {
    constexpr int TB = FB_TB2;
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t start = pstart[p], total = pstart[p + 1] - start;
    if (total <= (uint32_t)FB_STAGE || lng[FB_LONG_IDX + p] != FB_LONG_NONE) return;   // (fits the stage / split over workgroups)
    SORT_PRIO();
    __shared__ uint32_t h[FB_KEYS2 + 8];
    __shared__ uint32_t ws[17];
    __shared__ uint32_t rstart[FB_MAX_TILES];
    __shared__ uint32_t rpre[FB_MAX_TILES + 8];
    for (uint32_t b = tid; b <= FB_KEYS2; b += TB) h[b] = 0;
    for (uint32_t t = tid; t < ntile; t += TB) {
        const uint32_t* tt = toff + (size_t)t * (FB_NPART + 1) + p;
        const uint32_t a = tt[0];
        rstart[t] = a;
        rpre[t] = tt[1] - a;
    }
    fb_block_scan(rpre, ntile, ws);
    const bool ext = p >= FB_PPW;
    const uint32_t key0 = ext ? FB_NB + ((p - FB_PPW) << FB_XSUB_BITS) : p << FB_SUB_BITS;
    const uint32_t nslot = ext ? FB_XSB : FB_SB, lv = ext ? 1u : (uint32_t)FB_W, nkeys = nslot * lv;
    // Two shapes of "too long": long runs (one digit of small scalars: a tile's 2048 keys of a level in ONE partition) go
    // tile-major, a wave per run; many short runs (a partition a little over the stage: 30 k keys as 2048 runs of 15) would
    // leave 15 of a wave's 512 key slots busy that way and go key-major instead, each key finding its tile by an eleven-step
    // search over the runs' prefix sums.
    const bool by_tile = total / ntile >= 48u;
    auto key_of = [&](uint32_t it) -> uint32_t { return ext ? it >> 16 : (it >> 16) * FB_W + ((it >> 11) & 15u); };
    auto locate = [&](uint32_t k, uint32_t& t) -> const uint32_t* {
        uint32_t lo = 0;
#pragma unroll
        for (int sft = 10; sft >= 0; --sft) {
            const uint32_t mid = lo + (1u << sft);
            const uint32_t v = rpre[mid < (uint32_t)FB_MAX_TILES ? mid : (uint32_t)FB_MAX_TILES];
            lo = (mid < ntile && v <= k) ? mid : lo;
        }
        t = lo;
        return items + (size_t)lo * FB_KEYS1 + rstart[lo] + (k - rpre[lo]);
    };
    constexpr int MB = 4;
    const uint32_t rounds = (total + MB * TB - 1) / (MB * TB);
    if (by_tile) fb_long_pass<false>(items, (const lds_u32*)rstart, (const lds_u32*)rpre, (lds_u32*)h, nullptr, ntile, n_level, ext);
    else
        for (uint32_t r = 0; r < rounds; ++r) {
            uint32_t it[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const uint32_t k = (r * MB + j) * TB + tid;
                uint32_t t;
                it[j] = k < total ? *locate(k, t) : 0u;
            }
#pragma unroll
            for (int j = 0; j < MB; ++j) dm_wave_agg_add((lds_u32*)h, key_of(it[j]), (r * MB + j) * TB + tid < total);
        }
    __syncthreads();
    for (uint32_t b = tid; b < nslot; b += TB) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < lv; ++w) sum += h[b * lv + w];
        hist[key0 + b] = sum;
    }
    fb_block_scan(h, nkeys, ws);
    for (uint32_t b = tid; b < nslot; b += TB) offs[key0 + b] = start + h[b * lv];
    __syncthreads();
    if (by_tile) fb_long_pass<true>(items, (const lds_u32*)rstart, (const lds_u32*)rpre, (lds_u32*)h, entries + start, ntile, n_level, ext);
    else
        for (uint32_t r = 0; r < rounds; ++r) {
            uint32_t it[MB], tt[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const uint32_t k = (r * MB + j) * TB + tid;
                tt[j] = 0;
                it[j] = k < total ? *locate(k, tt[j]) : 0u;
            }
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const uint32_t k = (r * MB + j) * TB + tid;
                const uint32_t at = dm_wave_agg_add((lds_u32*)h, key_of(it[j]), k < total);
                if (k < total) entries[start + at] = ((it[j] >> 15) & 1u) << 31 | (((it[j] >> 11) & 15u) * n_level + tt[j] * FB_T + (it[j] & 2047u));
            }
        }
}

}  // namespace h2agg
