// Pippenger bucket MSM over BN254 G1 for gfx950.
//
// Stands behind MockEccChip::multi_exp / ArithEccChip::multi_exp
//   halo2-snark-aggregator-api/src/mock/arith/ecc.rs:106-129, halo2-snark-aggregator-api/src/arith/ecc.rs:42-60
// which compute sum_i s_i * P_i as n independent 254-step double-and-add products.  The group element
// is the same; the schedule is built for 64-lane wavefronts:
//
//   digits      each lane recodes one canonical scalar into W signed c-bit digits (carry chain in
//               registers, the 256-bit value is shifted down c bits per window), coalesced 32-B reads
//   sort        two-level LDS partition of the (window, |digit|) keys (sort_kernels.hpp)
//               -> hist[W * 2^(c-1)], offs[], entries[n * W] (point index | sign << 31), order[]
//   accumulate  one lane per bucket (longest first) walks its run: 64-B gathers of Montgomery affine
//               bases, XYZZ mixed adds; buckets longer than `big` are handed to a workgroup each
//   reduce      per window sum_j (j+1) * B_j by running sums over segments of `seg` buckets, segment
//               offsets folded in with a <= 15-bit double-and-add, then one workgroup per window
//   final       Horner over the W window sums (c doublings each), canonical Jacobian out
//
// Point order inside a bucket is whatever the scatter's atomics produced; the result does not depend on
// it because the arithmetic is exact.
#pragma once
#include "sort_kernels.hpp"
#include "fb_sort_kernels.hpp"

namespace h2agg {

struct MsmPlan {
    int c;          // window bits
    int W;          // number of windows, W*c >= 255
    uint32_t NB;    // buckets per window = 2^(c-1)
    uint32_t NBT;   // W * NB
    uint32_t seg;   // buckets per reduce segment (power of two, <= NB)
    uint32_t spw;   // segments per window = NB / seg
    uint32_t big;   // bucket length above which a workgroup takes the bucket
    bool glv;       // scalars split by the endomorphism: W*c >= 128, two insertions per (point, window)
};

// ------------------------------------------------------------------ bucket accumulation
// phi(P) = (beta * x, y): beta * x is computed ONCE per base (k_bases_endo_x, 32 B / point beside the table) instead of
// once per bucket entry — with the multiplication inside the gather, every wave that holds one endo entry paid 223
// instructions on all its lanes, 8 % of the accumulation.  The identity (0, 0) maps to itself.
FP_INLINE G1Affine msm_gather(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ endo_x, uint32_t e) {
    const size_t idx = e & ENT_IDX;
    G1Affine p;
    p.x = fp_load<FqParams>((e & ENT_ENDO) ? endo_x + 32 * idx : bases + 64 * idx);
    p.y = fp_load<FqParams>(bases + 64 * idx + 32);
    return (e & ENT_NEG) ? affine_neg(p) : p;
}
__global__ void __launch_bounds__(BLOCK) k_bases_endo_x(const uint8_t* __restrict__ bases, size_t n,
                                                        uint8_t* __restrict__ endo_x) {
    Fq beta;
#pragma unroll
    for (int i = 0; i < NL; ++i) beta.l[i] = GlvConst::BETA_MONT[i];
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK)
        fp_store<FqParams>(endo_x + 32 * i, FQ_MUL(fp_load<FqParams>(bases + 64 * i), beta));   // 2*1/169 + 1 -> [2]
}

// fixed-base tables: level w holds 2^(c*w) * P_i (Montgomery affine) for every base, so an MSM over the table needs no
// per-window bucket sets and no doubling chain — every digit of every scalar lands in ONE bucket set, and the width can
// exceed 16 bits because the bucket reduction is paid once, not once per window.  One level from the previous one:
__global__ void __launch_bounds__(BLOCK) k_bases_shift(const uint8_t* __restrict__ in, size_t n, int c,
                                                       uint8_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const G1Affine a = affine_load(in + 64 * i);
        G1XYZZ p = G1XYZZ::identity();
        xyzz_add_affine(p, a);
#pragma unroll 1
        for (int k = 0; k < c; ++k) p = xyzz_double(p);
        affine_store(out + 64 * i, affine_from_xyzz(p));
    }
}

// Buckets longer than `big` are cut into chunks of BIG_CHUNK entries, one workgroup per chunk (a narrow top
// window or skewed scalars can put a large share of all points into a handful of buckets — a 2-bit top
// window at c = 14 holds n/4 points per bucket).  big_list: 3 words per chunk slot {key, chunk, nchunks};
// big_keys: 3 words per multi-chunk bucket {key, first slot, nchunks}.
constexpr uint32_t BIG_CHUNK = 512;        // smallest chunk (a one-wave workgroup walks it: 8 additions per lane)
constexpr uint32_t BIG_CHUNK_MAX = 4096;   // largest: 64 additions per lane amortise the wave's LDS tree (6 levels)
// chunk size of an over-long bucket: ~256 chunks per bucket (a bucket holding a million entries — all scalars equal — is
// then 256 one-wave workgroups of 64 additions per lane, not 2048 of 8 additions + a tree each)
FP_INLINE uint32_t big_chunk_size(uint32_t len) {
    uint32_t cs = len / 256;
    cs = cs < BIG_CHUNK ? BIG_CHUNK : (cs > BIG_CHUNK_MAX ? BIG_CHUNK_MAX : cs);
    return cs;
}

// CHAIN (slices of one MSM sharing one bucket set, h2agg.hip `msm_run` chain modes): 0 = a whole MSM; 1 = first slice (like 0,
// but an over-long bucket's unused slice slots are set to the identity so that later slices can resume from them); 2 = later
// slice: every lane starts from the sum its slot already holds and leaves it alone when the slice has nothing for it.
template <int CHAIN>
__global__ void __launch_bounds__(BLOCK) k_msm_accumulate(const uint8_t* __restrict__ bases,
                                                          const uint8_t* __restrict__ endo_x,
                                                          const uint32_t* __restrict__ entries,
                                                          const uint32_t* __restrict__ offs,
                                                          const uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ order, uint32_t nbt, uint32_t big,
                                                          uint32_t lpb /* lanes per bucket: 1, 2 or 4 */,
                                                          uint8_t* __restrict__ buckets, uint32_t* __restrict__ big_list,
                                                          uint32_t* __restrict__ big_keys,
                                                          uint32_t* __restrict__ counters /* [0] chunks, [1] keys */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbt * lpb) return;
    // buckets sorted by length, longest first: a wave's lanes finish together.  With lpb > 1 each bucket's run
    // is cut into lpb slices handled by adjacent lanes (more, shorter waves: fills the 3072 wave slots when
    // there are few buckets — GLV halves their number); the slices' partial sums go to buckets[key*lpb + part]
    // and are folded by k_msm_bucket_combine.
    const uint32_t part = t % lpb;
    const uint32_t key = order ? order[t / lpb] : t / lpb;   // (small MSMs skip the ordering pass)
    const uint32_t len = hist[key];
    if (len > big) {
        if (part != 0) {
            if (CHAIN == 1) xyzz_store(buckets + XYZZ_BYTES * ((size_t)key * lpb + part), G1XYZZ::identity());
            return;
        }
        const uint32_t cs = big_chunk_size(len);
        const uint32_t nch = (len + cs - 1) / cs;
        const uint32_t base = atomicAdd(&counters[0], nch);
        for (uint32_t j = 0; j < nch; ++j) {
            big_list[3 * (base + j)] = key;
            big_list[3 * (base + j) + 1] = j;
            big_list[3 * (base + j) + 2] = nch;
        }
        if (nch > 1) {
            const uint32_t k = atomicAdd(&counters[1], 1u);
            big_keys[3 * k] = key;
            big_keys[3 * k + 1] = base;
            big_keys[3 * k + 2] = nch;
        }
        return;
    }
    const uint32_t lo = (uint32_t)(((uint64_t)len * part) / lpb), hi = (uint32_t)(((uint64_t)len * (part + 1)) / lpb);
    const uint32_t* run = entries + offs[key];
    G1XYZZ acc = G1XYZZ::identity();
    uint32_t k0 = lo;
    bool fresh = true;
    if (CHAIN == 2) {
        if (hi == lo) return;
        acc = xyzz_load(buckets + XYZZ_BYTES * ((size_t)key * lpb + part));
        fresh = acc.is_identity();
    }
    // The first point of a bucket is a copy and the second an affine + affine addition (4M + 2S): both outside the loop,
    // which then only ever sees the 8M + 2S mixed addition.  (Identity bases are skipped; a lane that meets some simply
    // enters the loop at its own k0.)
    G1Affine first;
    bool have_first = false;
    while (fresh && k0 < hi && !have_first) {
        first = msm_gather(bases, endo_x, run[k0++]);
        have_first = !first.is_identity();
    }
    G1Affine second;
    bool have_second = false;
    while (have_first && k0 < hi && !have_second) {
        second = msm_gather(bases, endo_x, run[k0++]);
        have_second = !second.is_identity();
    }
    if (have_second) acc = xyzz_add_affine_affine(first, second);
    else if (have_first) acc = G1XYZZ::from_affine(first);
    if (hi > k0) {
        // two-deep software pipeline: the base of entry k + 1 is gathered, and the INDEX of entry k + 2 loaded, under the
        // addition of entry k — the index load used to sit in front of its gather with a full `s_waitcnt vmcnt(0)`
        const uint32_t lo = k0;
        G1Affine nxt = msm_gather(bases, endo_x, run[lo]);
        uint32_t e_after = lo + 1 < hi ? run[lo + 1] : 0u;
#pragma unroll 1
        for (uint32_t k = lo; k < hi; ++k) {
            G1Affine cur = nxt;
            if (k + 1 < hi) {
                nxt = msm_gather(bases, endo_x, e_after);
                e_after = k + 2 < hi ? run[k + 2] : 0u;
            }
            xyzz_add_affine(acc, cur);
        }
    }
    xyzz_store(buckets + XYZZ_BYTES * ((size_t)key * lpb + part), acc);
}

// ---- the lean accumulation: 128 VGPRs, no scratch, four waves per SIMD ---------------------------------------------
// k_msm_accumulate above keeps ~166 VGPRs alive (LLVM interleaves the independent products of a mixed addition), which
// pins it at three waves per SIMD with nothing beside it.  Here every Montgomery product is one opaque inline-asm block
// with hand-picked temporaries (fp_asm.inc, generated by tools/gen_fp_asm.py) and the insertion is STRAIGHT-LINE: the
// exceptional cases of the group law — an identity operand, q = +-acc — are not decided in the loop at all; a lane that
// meets one (a one-limb filter that cannot miss a multiple of p) stores the sum it has, appends (slot, entry index) to
// fix_list and leaves; k_msm_accumulate_fix finishes those buckets with the general formulas.  For uniform scalars the
// list holds the filter's false positives: ~10 * 2^-58 per insertion (two limbs), i.e. it is empty.
struct RawPt { uint4 a, b, c, d; };   // x (a, b) || y (c, d) as loaded: 16 registers while the gather is in flight
FP_INLINE RawPt raw_gather(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ endo_x, uint32_t e) {
    const size_t idx = e & ENT_IDX;
    const uint4* px = reinterpret_cast<const uint4*>((e & ENT_ENDO) ? endo_x + 32 * idx : bases + 64 * idx);
    const uint4* py = reinterpret_cast<const uint4*>(bases + 64 * idx + 32);
    RawPt r;
    r.a = px[0]; r.b = px[1]; r.c = py[0]; r.d = py[1];
    return r;
}
FP_INLINE G1Affine raw_unpack(const RawPt& r, uint32_t e) {
    G1Affine p;
    { uint32_t w[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w}; p.x = fp_unpack<FqParams>(w); }
    { uint32_t w[8] = {r.c.x, r.c.y, r.c.z, r.c.w, r.d.x, r.d.y, r.d.z, r.d.w}; p.y = fp_unpack<FqParams>(w); }
    return (e & ENT_NEG) ? affine_neg(p) : p;
}
// acc += q (madd-2008-s, same bounds as xyzz_add_affine).  Returns false, acc untouched, when the general formula is
// needed.  issue_next() starts the gather of the following entry (16 registers while in flight) at the point of the formula where
// the fewest values are alive: behind (PPP, Q), with the last four products still to come.  (Measured alternative: the gather as
// LDS-DMA, global_load_lds_dwordx4 into a per-wave stage, a whole addition ahead and no registers at all — 1 102 us per 2^20-point
// launch against 1 076: the lead is not what bounds the kernel.)
// DUAL: independent products as two chains in lock step (no dependent back-to-back multiply-adds: what three waves per SIMD
// need); otherwise one chain per product (fewest temporaries; four waves per SIMD hide the dependent issue).
template <bool DUAL, class NextF>
FP_INLINE bool xyzz_add_affine_lean(G1XYZZ& acc, G1Affine q, NextF&& issue_next) {
    bool ok = !q.is_identity() && !acc.is_identity();
    if (DUAL) {
        fpa_mul_dual_ip<FqParams>(q.x, acc.zz, q.y, acc.zzz);   // U2 [2], S2 [2]
    } else {
        fpa_mul_ip<FqParams>(q.x, acc.zz);
        fpa_mul_ip<FqParams>(q.y, acc.zzz);
    }
    Fq p = FQ_SUB(8, q.x, acc.x);                        // [10]
    Fq r = FQ_SUB(4, q.y, acc.y);                        // [6]
    ok = ok && !fp_maybe_zero_mod2<10, FqParams>(p);
    if (!ok) return false;
    Fq pp, rr;
    if (DUAL) {
        fpa_sqr_dual<FqParams>(pp, rr, p, r);            // 100 -> [2], 36 -> [2]
        fpa_mul_dual_ip<FqParams>(p, pp, acc.x, pp);     // PPP 20 -> [2], Q 16 -> [2]
    } else {
        fpa_sqr<FqParams>(pp, p);
        fpa_sqr<FqParams>(rr, r);
        fpa_mul_ip<FqParams>(p, pp);
        fpa_mul_ip<FqParams>(acc.x, pp);
    }
    Fq x3 = fp_sub_sub2<6, FqParams>(rr, p, acc.x);      // PPP + 2Q [6] -> [8]
    Fq d = FQ_SUB(8, acc.x, x3);                         // Q - X3 [10]
    acc.x = x3;
    Fq ny = fp_neg<4, FqParams>(acc.y);                  // [4]
    if (DUAL) {
        fpa_mul_dual_ip<FqParams>(acc.zz, pp, acc.zzz, p);   // [2], [2]
        issue_next();
        fpa_mul2_ip<FqParams>(ny, p, r, d);              // (4p - Y1)*PPP + R*(Q - X3): (4*2 + 6*10)/169 + 1 -> [2]
    } else {
        fpa_mul_ip<FqParams>(acc.zz, pp);
        fpa_mul_ip<FqParams>(acc.zzz, p);
        fpa_mul2_ip1<FqParams>(ny, p, r, d);
    }
    acc.y = ny;
    return true;
}
// a + q for two affine points (mmadd-2008-s, 4M + 2S): the second point of a bucket.  Same contract as above.
template <bool DUAL>
FP_INLINE bool xyzz_add_affine_affine_lean(G1XYZZ& o, const G1Affine& a, const G1Affine& q) {
    Fq p = FQ_SUB(2, q.x, a.x);                          // [4]
    Fq r = FQ_SUB(2, q.y, a.y);                          // [4]
    if (q.is_identity() || fp_maybe_zero_mod2<4, FqParams>(p)) return false;
    Fq pp, rr;
    Fq qq = a.x;
    if (DUAL) {
        fpa_sqr_dual<FqParams>(pp, rr, p, r);            // 16 -> [2], [2]
        fpa_mul_dual_ip<FqParams>(p, pp, qq, pp);        // PPP [2], Q [2]
    } else {
        fpa_sqr<FqParams>(pp, p);
        fpa_sqr<FqParams>(rr, r);
        fpa_mul_ip<FqParams>(p, pp);
        fpa_mul_ip<FqParams>(qq, pp);
    }
    o.x = fp_sub_sub2<6, FqParams>(rr, p, qq);           // [8]
    Fq d = FQ_SUB(8, qq, o.x);                           // [10]
    Fq ny = fp_neg<2, FqParams>(a.y);                    // [2]
    if (DUAL) fpa_mul2_ip<FqParams>(ny, p, r, d);        // (2*2 + 4*10)/169 + 1 -> [2]
    else fpa_mul2_ip1<FqParams>(ny, p, r, d);
    o.y = ny;
    o.zz = pp;
    o.zzz = p;
    return true;
}

// Same contract as k_msm_accumulate<CHAIN>; counters[2] counts fix_list's {slot, first unfinished entry} pairs.
template <int CHAIN, bool DUAL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_msm_accumulate_lean(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ endo_x,
                      const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offs,
                      const uint32_t* __restrict__ hist, const uint32_t* __restrict__ order, uint32_t nbt, uint32_t big,
                      uint32_t lpb, uint8_t* __restrict__ buckets, uint32_t* __restrict__ big_list,
                      uint32_t* __restrict__ big_keys, uint32_t* __restrict__ counters, uint32_t* __restrict__ fix_list) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbt * lpb) return;
    const uint32_t part = t % lpb;
    const uint32_t key = order ? order[t / lpb] : t / lpb;
    const uint32_t len = hist[key];
    const uint32_t slot = key * lpb + part;
    if (len > big) {   // over-long bucket: chunk list for k_msm_accumulate_big (as in k_msm_accumulate)
        if (part != 0) {
            if (CHAIN == 1) xyzz_store(buckets + XYZZ_BYTES * (size_t)slot, G1XYZZ::identity());
            return;
        }
        const uint32_t cs = big_chunk_size(len);
        const uint32_t nch = (len + cs - 1) / cs;
        const uint32_t base = atomicAdd(&counters[0], nch);
        for (uint32_t j = 0; j < nch; ++j) {
            big_list[3 * (base + j)] = key;
            big_list[3 * (base + j) + 1] = j;
            big_list[3 * (base + j) + 2] = nch;
        }
        if (nch > 1) {
            const uint32_t kk = atomicAdd(&counters[1], 1u);
            big_keys[3 * kk] = key;
            big_keys[3 * kk + 1] = base;
            big_keys[3 * kk + 2] = nch;
        }
        return;
    }
    const uint32_t lo = (uint32_t)(((uint64_t)len * part) / lpb), hi = (uint32_t)(((uint64_t)len * (part + 1)) / lpb);
    // (register budget: the run is addressed as entries[roff + k] off the kernel argument, not through a per-lane 64-bit pointer)
    const uint32_t roff = offs[key];
    auto run = [&](uint32_t i) { return entries[roff + i]; };
    G1XYZZ acc = G1XYZZ::identity();
    if (CHAIN == 2) {
        if (hi == lo) return;
        acc = xyzz_load(buckets + XYZZ_BYTES * (size_t)slot);
    }
    uint32_t k = lo;
    if (hi > lo) {
        uint32_t e_cur = run(lo);
        RawPt nxt = raw_gather(bases, endo_x, e_cur);
        uint32_t e_nxt = run(lo + 1 < hi ? lo + 1 : hi - 1);
        auto advance = [&](uint32_t k_now) {   // gather entry k_now + 1 (past the end: the last entry again, unused), fetch the index of entry k_now + 2
            e_cur = e_nxt;
            nxt = raw_gather(bases, endo_x, e_nxt);
            e_nxt = run(k_now + 2 < hi ? k_now + 2 : hi - 1);
        };
        bool go = true;
        if (acc.is_identity()) {   // the first point of a bucket is a copy, the second an affine + affine addition (4M + 2S)
            const G1Affine first = raw_unpack(nxt, e_cur);
            go = !first.is_identity();
            if (go) {
                advance(lo);
                k = lo + 1;
                acc = G1XYZZ::from_affine(first);
                if (k < hi) {
                    const G1Affine second = raw_unpack(nxt, e_cur);
                    advance(lo + 1);
                    go = xyzz_add_affine_affine_lean<DUAL>(acc, first, second);
                    if (go) k = lo + 2;
                }
            }
        }
        if (go) {
#pragma unroll 1
            for (; k < hi; ++k) {
                const G1Affine cur = raw_unpack(nxt, e_cur);
                if (!xyzz_add_affine_lean<DUAL>(acc, cur, [&]() { advance(k); })) break;
            }
        }
        if (k < hi) {
            const uint32_t f = atomicAdd(&counters[2], 1u);
            fix_list[2 * f] = slot;
            fix_list[2 * f + 1] = k;
        }
    }
    uint32_t oslot = slot;
    asm volatile("" : "+v"(oslot));   // (keeps the 64-bit store address out of the loop's live set)
    xyzz_store(buckets + XYZZ_BYTES * (size_t)oslot, acc);
}
// the buckets the lean kernel left unfinished, with the general formulas: one lane per list entry, from entry k on
__global__ void __launch_bounds__(64) k_msm_accumulate_fix(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ endo_x,
                                                           const uint32_t* __restrict__ entries,
                                                           const uint32_t* __restrict__ offs,
                                                           const uint32_t* __restrict__ hist, uint32_t lpb,
                                                           uint8_t* __restrict__ buckets,
                                                           const uint32_t* __restrict__ counters,
                                                           const uint32_t* __restrict__ fix_list) {
    const uint32_t nfix = counters[2];
    for (uint32_t f = blockIdx.x * 64 + threadIdx.x; f < nfix; f += gridDim.x * 64) {
        const uint32_t slot = fix_list[2 * f], k0 = fix_list[2 * f + 1];
        const uint32_t key = slot / lpb, part = slot - key * lpb;
        const uint32_t len = hist[key];
        const uint32_t hi = (uint32_t)(((uint64_t)len * (part + 1)) / lpb);
        const uint32_t* run = entries + offs[key];
        G1XYZZ acc = xyzz_load(buckets + XYZZ_BYTES * (size_t)slot);
#pragma unroll 1
        for (uint32_t k = k0; k < hi; ++k) xyzz_add_affine(acc, msm_gather(bases, endo_x, run[k]));
        xyzz_store(buckets + XYZZ_BYTES * (size_t)slot, acc);
    }
}

// buckets[key] = sum of the lpb slice sums written by k_msm_accumulate (skipped for over-long buckets, whose
// sum the chunk path writes to slot key*lpb directly)
__global__ void __launch_bounds__(BLOCK) k_msm_bucket_combine(const uint8_t* __restrict__ parts,
                                                              const uint32_t* __restrict__ hist, uint32_t nbt,
                                                              uint32_t big, uint32_t lpb, uint8_t* __restrict__ buckets) {
    const uint32_t key = blockIdx.x * BLOCK + threadIdx.x;
    if (key >= nbt) return;
    G1XYZZ acc = xyzz_load(parts + XYZZ_BYTES * (size_t)key * lpb);
    if (hist[key] <= big) {
#pragma unroll 1
        for (uint32_t j = 1; j < lpb; ++j) acc = xyzz_add(acc, xyzz_load(parts + XYZZ_BYTES * ((size_t)key * lpb + j)));
    }
    xyzz_store(buckets + XYZZ_BYTES * (size_t)key, acc);
}

// Sum of one XYZZ point per lane over a ONE-WAVE workgroup (result valid in lane 0); lds: XYZZ_WORDS * 64 words.
// The over-long-bucket kernels run as one-wave workgroups on purpose: they sit at the head of the tail, beside the next
// MSM's accumulation, and a 256-thread workgroup of ~190-VGPR waves only finds room when that kernel is nearly over
// (it held the whole tail back by 0.9 ms although its lists are empty for uniform scalars); a single wave slips into the
// first slot a retiring accumulate wave leaves.
__device__ __noinline__ G1XYZZ wave_sum_xyzz(G1XYZZ v, uint32_t* lds) {
    const int lane = threadIdx.x;
    auto put = [&](int slot, const G1XYZZ& p) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            lds[i * 64 + slot] = p.x.l[i];
            lds[(NL + i) * 64 + slot] = p.y.l[i];
            lds[(2 * NL + i) * 64 + slot] = p.zz.l[i];
            lds[(3 * NL + i) * 64 + slot] = p.zzz.l[i];
        }
    };
    put(lane, v);
    __syncthreads();
#pragma unroll 1
    for (int s2 = 32; s2 >= 1; s2 >>= 1) {
        if (lane < s2) {
            G1XYZZ o;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                o.x.l[i] = lds[i * 64 + lane + s2];
                o.y.l[i] = lds[(NL + i) * 64 + lane + s2];
                o.zz.l[i] = lds[(2 * NL + i) * 64 + lane + s2];
                o.zzz.l[i] = lds[(3 * NL + i) * 64 + lane + s2];
            }
            v = xyzz_add(v, o);
            put(lane, v);
        }
        __syncthreads();
    }
    return v;
}

// one (one-wave) workgroup per chunk of an over-long bucket
__global__ void __launch_bounds__(64) k_msm_accumulate_big(const uint8_t* __restrict__ bases,
                                                           const uint8_t* __restrict__ endo_x,
                                                           const uint32_t* __restrict__ entries,
                                                           const uint32_t* __restrict__ offs,
                                                           const uint32_t* __restrict__ hist,
                                                           uint8_t* __restrict__ buckets, uint32_t stride /* lpb */,
                                                           uint8_t* __restrict__ big_part,
                                                           const uint32_t* __restrict__ big_list,
                                                           const uint32_t* __restrict__ counters,
                                                           uint32_t resume /* later slice of a chain: add to what the slot holds */) {
    __shared__ uint32_t lds[XYZZ_WORDS * 64];
    const uint32_t nslots = counters[0];
    for (uint32_t b = blockIdx.x; b < nslots; b += gridDim.x) {
        const uint32_t key = big_list[3 * b], chunk = big_list[3 * b + 1], nch = big_list[3 * b + 2];
        const uint32_t len = hist[key];
        const uint32_t cs = big_chunk_size(len);
        const uint32_t lo = chunk * cs, hi = (lo + cs < len) ? lo + cs : len;
        const uint32_t* run = entries + offs[key];
        G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
        for (uint32_t k = lo + threadIdx.x; k < hi; k += 64) xyzz_add_affine(acc, msm_gather(bases, endo_x, run[k]));
        G1XYZZ tot = wave_sum_xyzz(acc, lds);
        if (threadIdx.x == 0) {
            if (nch == 1 && resume) tot = xyzz_add(tot, xyzz_load(buckets + XYZZ_BYTES * (size_t)key * stride));
            xyzz_store(nch == 1 ? buckets + XYZZ_BYTES * (size_t)key * stride : big_part + XYZZ_BYTES * (size_t)b, tot);
        }
        __syncthreads();
    }
}
// sum the chunk partials of every multi-chunk bucket
__global__ void __launch_bounds__(64) k_msm_big_combine(const uint8_t* __restrict__ big_part,
                                                        const uint32_t* __restrict__ big_keys,
                                                        const uint32_t* __restrict__ counters,
                                                        uint8_t* __restrict__ buckets, uint32_t stride /* lpb */, uint32_t resume) {
    __shared__ uint32_t lds[XYZZ_WORDS * 64];
    const uint32_t nkeys = counters[1];
    for (uint32_t k = blockIdx.x; k < nkeys; k += gridDim.x) {
        const uint32_t key = big_keys[3 * k], base = big_keys[3 * k + 1], nch = big_keys[3 * k + 2];
        G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
        for (uint32_t j = threadIdx.x; j < nch; j += 64)
            acc = xyzz_add(acc, xyzz_load(big_part + XYZZ_BYTES * (size_t)(base + j)));
        G1XYZZ tot = wave_sum_xyzz(acc, lds);
        if (threadIdx.x == 0) {
            if (resume) tot = xyzz_add(tot, xyzz_load(buckets + XYZZ_BYTES * (size_t)key * stride));
            xyzz_store(buckets + XYZZ_BYTES * (size_t)key * stride, tot);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ bucket reduction
// k * P for a small k (< 2^16), MSB-first
__device__ __noinline__ G1XYZZ xyzz_mul_small(const G1XYZZ& p, uint32_t k) {
    G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
    for (int bit = 31 - __clz((int)(k | 1u)); bit >= 0; --bit) {
        acc = xyzz_double(acc);
        if ((k >> bit) & 1u) acc = xyzz_add(acc, p);
    }
    return acc;
}

// segsum[w][s] = sum_{j in segment s} (j+1) * B[w][j]
__global__ void __launch_bounds__(BLOCK) k_msm_reduce_segments(const uint8_t* __restrict__ buckets, uint32_t NB,
                                                               uint32_t seg, uint32_t spw, uint32_t total,
                                                               uint8_t* __restrict__ segsum) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= total) return;
    const uint32_t w = t / spw, sidx = t - w * spw;
    const uint32_t a = sidx * seg;
    const uint8_t* B = buckets + XYZZ_BYTES * ((size_t)w * NB + a);
    G1XYZZ running = G1XYZZ::identity(), acc = G1XYZZ::identity();
#pragma unroll 1
    for (int j = (int)seg - 1; j >= 0; --j) {
        running = xyzz_add(running, xyzz_load(B + XYZZ_BYTES * (size_t)j));
        acc = xyzz_add(acc, running);
    }
    // sum (a + jj + 1) B = acc + a * running
    if (a != 0 && !running.is_identity()) acc = xyzz_add(acc, xyzz_mul_small(running, a));
    xyzz_store(segsum + XYZZ_BYTES * (size_t)t, acc);
}

// wsum[w] = sum_s segsum[w][s]; one workgroup per window
__global__ void __launch_bounds__(BLOCK) k_msm_window_sum(const uint8_t* __restrict__ segsum, uint32_t spw,
                                                          uint8_t* __restrict__ wsum) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    const uint32_t w = blockIdx.x;
    G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
    for (uint32_t s = threadIdx.x; s < spw; s += BLOCK)
        acc = xyzz_add(acc, xyzz_load(segsum + XYZZ_BYTES * ((size_t)w * spw + s)));
    G1XYZZ tot = block_sum_xyzz(acc, lds);
    if (threadIdx.x == 0) xyzz_store(wsum + XYZZ_BYTES * (size_t)w, tot);
}

// ---- two-dimensional bucket reduction (16-bit windows: 2^15 buckets = 256 rows x 128 columns) -----------------------
// With the tails under the next MSM's bulk the whole step is VALU-issue-bound, and the segment kernels above are 124 M
// wave-instructions per 2^20-point MSM against 548 M for the bucket accumulation (skipping them: 1.44 -> 1.20 ms per step).
// They pay for short chains: 4 lanes per chain at 16 lane-products per 14-product addition, plus a double-and-add by the
// segment offset per segment.  Writing the bucket index as b = 128 h + l,
//     sum_b (b + 1) B_b  =  128 * sum_h h R_h  +  sum_l (l + 1) C_l,      R_h = sum_l B[h][l],   C_l = sum_h B[h][l],
// turns the bulk of the work into PLAIN sums — two additions per bucket as before, but independent ones: one lane each,
// every lane busy, chains of 16 — and leaves two weighted sums over 256 and 128 points per window for one workgroup.
// The grid is 256 rows x 2^LC columns: LC = 7 for 16-bit windows (2^15 buckets), LC = 8 for 17-bit windows (2^16 buckets).
constexpr int R2D_ROWS = 256, R2D_L = 16;
template <int LC>
struct R2D {
    static constexpr int LCOLS = LC, COLS = 1 << LC;
    static constexpr int RPARTS = COLS / R2D_L, CPARTS = R2D_ROWS / R2D_L;               // LC = 7: 8 row parts, 16 column parts
    static constexpr int ROW_THREADS = R2D_ROWS * RPARTS, COL_THREADS = COLS * CPARTS;   // LC = 7: 2048 + 2048 per window
    static constexpr int THREADS = ROW_THREADS + COL_THREADS;
};
constexpr int R2D_COLS = R2D<7>::COLS, R2D_THREADS = R2D<7>::THREADS;   // (the 16-bit grid: what the host sizes by default)

// 64 XYZZ records, one per lane, fetched by the wave as ONE 9216-byte block: slot g = 64 q + lane of the block is piece
// g % 9 of lane g / 9's record, so consecutive lanes ask for consecutive 16-byte pieces (whole lines, each fetched once —
// a lane reading its own record piece by piece touches 64 different lines per instruction and uses a quarter of each),
// and CDNA4's LDS-DMA load (global_load_lds_dwordx4) drops them straight into LDS: no staging registers, the block for the
// next addition is in flight while this one is computed.
FP_INLINE void r2d_fetch(const uint8_t* __restrict__ B, const uint32_t (&off)[9], uint32_t step_bytes, uint32_t* lds_block) {
#pragma unroll
    for (int q = 0; q < 9; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + off[q] + step_bytes),
                                         (__attribute__((address_space(3))) void*)(lds_block + 256 * q), 16, 0, 0);
}
FP_INLINE G1XYZZ r2d_own(const uint32_t* lds_block, int lane) {
    const uint4* q = reinterpret_cast<const uint4*>(lds_block) + 9 * lane;
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint4 v = q[i];
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

// parts[w * 4096 + t]: t < 2048: row h = t >> 3, part j = t & 7: sum of B[h][j + 8 i], i < 16
//                      t >= 2048: u = t - 2048, part j = u >> 7, column l = u & 127: sum of B[j + 16 i][l], i < 16
// One-wave workgroups; a wave is all row parts or all column parts, so the step between a lane's records is wave-uniform.
template <int LC>
__global__ void __launch_bounds__(64) k_msm_reduce2d_parts(const uint8_t* __restrict__ buckets, uint32_t total,
                                                           uint8_t* __restrict__ parts) {
    typedef R2D<LC> G;
    __shared__ __attribute__((aligned(16))) uint32_t blk[2][9 * 256];
    const int lane = threadIdx.x;
    // (one chain per lane.  Two chains per lane — half the one-wave workgroups, to leave the next MSM's sort more LDS — was tried
    // for the 256 x 256 grid and lost: the kernel then runs twice as long beside that sort, 1.22 -> 1.41 ms per step.)
    const uint32_t g = blockIdx.x * 64 + lane;
    {
    const uint32_t w = g / G::THREADS, t = g - w * G::THREADS;
    uint32_t base, stride;
    if (t < (uint32_t)G::ROW_THREADS) {            // row h = t / RPARTS, part j = t % RPARTS: B[h][j + RPARTS i], i < 16
        base = (t / G::RPARTS) * G::COLS + (t % G::RPARTS);
        stride = G::RPARTS;
    } else {                                       // part j = u >> LC, column l: B[j + 16 i][l], i < 16
        const uint32_t u = t - G::ROW_THREADS;
        base = (u >> G::LCOLS) * G::COLS + (u & (G::COLS - 1));
        stride = G::CPARTS * G::COLS;
    }
    const uint8_t* B = buckets + XYZZ_BYTES * (size_t)w * (R2D_ROWS * G::COLS);
    uint32_t off[9];   // byte offset (from B) of the piece this lane fetches in round q
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const uint32_t slot = 64 * q + lane, r = slot / 9, piece = slot - 9 * r;
        off[q] = (uint32_t)XYZZ_BYTES * (uint32_t)__shfl((int)base, (int)r, 64) + 16 * piece;
    }
    const uint32_t step = (uint32_t)XYZZ_BYTES * stride;
    r2d_fetch(B, off, 0, blk[0]);
    __syncthreads();
    G1XYZZ acc = r2d_own(blk[0], lane);
    r2d_fetch(B, off, step, blk[1]);
#pragma unroll 1
    for (int i = 1; i < R2D_L; ++i) {
        __syncthreads();
        const G1XYZZ cur = r2d_own(blk[i & 1], lane);
        __syncthreads();
        if (i + 1 < R2D_L) r2d_fetch(B, off, step * (i + 1), blk[(i + 1) & 1]);
        acc = xyzz_add_chains(acc, cur);
    }
    if (g < total) xyzz_store(parts + XYZZ_BYTES * (size_t)g, acc);
    }
}

constexpr int R2D_TB = 256;   // one wave per SIMD: every level of the scans below costs one addition's latency
FP_INLINE void r2d_put(uint32_t* lds, int slot, const G1XYZZ& p) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        lds[i * R2D_TB + slot] = p.x.l[i];
        lds[(NL + i) * R2D_TB + slot] = p.y.l[i];
        lds[(2 * NL + i) * R2D_TB + slot] = p.zz.l[i];
        lds[(3 * NL + i) * R2D_TB + slot] = p.zzz.l[i];
    }
}
FP_INLINE G1XYZZ r2d_get(const uint32_t* lds, int slot) {
    G1XYZZ p;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        p.x.l[i] = lds[i * R2D_TB + slot];
        p.y.l[i] = lds[(NL + i) * R2D_TB + slot];
        p.zz.l[i] = lds[(2 * NL + i) * R2D_TB + slot];
        p.zzz.l[i] = lds[(3 * NL + i) * R2D_TB + slot];
    }
    return p;
}

// Grid (windows, 2).  Workgroup (w, 0): 128 * sum_h h R_h over the 256 rows; (w, 1): sum_l (l + 1) C_l over the 128 columns
// (two threads per column).  sum_k k X_k over a ramp = sum of the suffix sums S_k = sum_{i >= k} X_i, k >= 1 (columns:
// weights l + 1, so S_0 counts too): a Hillis-Steele suffix scan and a tree, both log-depth, instead of a serial running
// sum.  The second workgroup of a window to finish adds the two halves into wsum[w] (ticket[w] counts arrivals and is left
// at zero again).
template <int LC>
__global__ void __launch_bounds__(R2D_TB) k_msm_reduce2d_window(const uint8_t* __restrict__ parts,
                                                                 uint8_t* __restrict__ halves /* [2 * windows] XYZZ */,
                                                                 uint32_t* __restrict__ ticket,
                                                                 uint8_t* __restrict__ wsum,
                                                                 uint8_t* __restrict__ tsum = nullptr /* [windows] XYZZ: plain sum of the window's buckets */) {
    typedef R2D<LC> G;
    static_assert(LC == 7 || LC == 8, "128 columns (two threads each) or 256 (one thread each) for 256 threads");
    __shared__ uint32_t lds[XYZZ_WORDS * R2D_TB];
    __shared__ uint32_t last_flag;
    const int tid = threadIdx.x;
    const uint32_t w = blockIdx.x;
    const bool rows = blockIdx.y == 0;
    const uint8_t* P = parts + XYZZ_BYTES * (size_t)w * G::THREADS;
    // element `pos` of the group lives in LDS slot pos * sl; 128 columns: slots 2l (the odd threads only feed the first level)
    constexpr bool two_per_col = LC == 7;
    const int sl = (rows || !two_per_col) ? 1 : 2;
    const int pos = (rows || !two_per_col) ? tid : (tid >> 1);
    const int n = rows ? R2D_ROWS : G::COLS;
    const bool owner = rows || !two_per_col || (tid & 1) == 0;
    const int lg = rows ? 8 : LC;   // log2(n)
    G1XYZZ v;
    {
        const uint8_t* src;
        size_t step;
        int cnt;
        if (rows) {
            src = P + XYZZ_BYTES * (size_t)(tid * G::RPARTS);
            step = XYZZ_BYTES;
            cnt = G::RPARTS;
        } else if (two_per_col) {
            src = P + XYZZ_BYTES * (size_t)(G::ROW_THREADS + (8 * (tid & 1)) * G::COLS + (tid >> 1));
            step = XYZZ_BYTES * (size_t)G::COLS;
            cnt = 8;
        } else {
            src = P + XYZZ_BYTES * (size_t)(G::ROW_THREADS + tid);
            step = XYZZ_BYTES * (size_t)G::COLS;
            cnt = G::CPARTS;
        }
        v = xyzz_load(src);
        G1XYZZ nxt = xyzz_load(src + step);
#pragma unroll 1
        for (int j = 1; j < cnt; ++j) {
            const G1XYZZ cur = nxt;
            if (j + 1 < cnt) nxt = xyzz_load(src + step * (j + 1));
            v = xyzz_add(v, cur);
        }
    }
    r2d_put(lds, tid, v);
    __syncthreads();
    // levels: [columns only: the two halves] | suffix scan, d = 1, 2, .. n/2 | S_0 := 0 for rows | tree, s = n/2 .. 1
    // ONE inlined addition serves them all (the operand's slot and the "active" predicate change per level)
    const int first = (rows || !two_per_col) ? 1 : 0;
#pragma unroll 1
    for (int lv = first; lv <= 2 * lg; ++lv) {
        int partner;
        bool act;
        if (lv == 0) {               // columns: even thread += odd thread
            partner = tid + 1;
            act = owner;
        } else if (lv <= lg) {       // scan
            const int d = 1 << (lv - 1);
            partner = (pos + d) * sl;
            act = owner && pos + d < n;
        } else {                     // tree
            const int s2 = n >> (lv - lg);
            partner = (pos + s2) * sl;
            act = owner && pos < s2;
            if (lv == lg + 1 && rows && tid == 0) {   // S_0 of the rows does not count (weight 0)
                if (tsum) xyzz_store(tsum + XYZZ_BYTES * (size_t)w, v);   // ... but it is the sum of all the window's buckets
                v = G1XYZZ::identity();
            }
        }
        G1XYZZ o;
        if (act) o = r2d_get(lds, partner);
        __syncthreads();
        if (act) v = xyzz_add(v, o);
        if (act || (lv == lg + 1 && rows && tid == 0)) r2d_put(lds, tid, v);
        __syncthreads();
    }
    if (tid == 0) {
        if (rows) {
#pragma unroll 1
            for (int k = 0; k < G::LCOLS; ++k) v = xyzz_double(v);
        }
        xyzz_store(halves + XYZZ_BYTES * (size_t)(2 * w + blockIdx.y), v);
        __threadfence();
        last_flag = atomicAdd(&ticket[w], 1u);
    }
    __syncthreads();
    if (last_flag == 1u && tid == 0) {
        __threadfence();
        const G1XYZZ other = xyzz_load(halves + XYZZ_BYTES * (size_t)(2 * w + (1 - blockIdx.y)));
        xyzz_store(wsum + XYZZ_BYTES * (size_t)w, xyzz_add(v, other));
        ticket[w] = 0;
    }
}

// ---- the one bucket set of an MSM over fixed-base levels at c = 20 (fb_sort_kernels.hpp): 2^19 buckets = 16 grids of 256 x 128
// sum_b (b + 1) B_b over b = 2^15 q + b'  =  sum_q [ sum_b' (b' + 1) B_{q, b'} ]  +  2^15 sum_q q T_q,   T_q = sum_b' B_{q, b'}:
// the two-dimensional reduction above over 16 pseudo-windows (k_msm_reduce2d_window also hands out T_q), then
// k_fb_wsum: out[0] = sum_q V_q, out[1] = sum_q q T_q, and k_msm_final_lp over those two "windows" of 15 bits.
constexpr int FB_R2D_WINDOWS = (int)(FB_NB / (uint32_t)(R2D_ROWS * R2D<7>::COLS));   // 16
// buckets[m - 1] += the sixteen parts of value m of the top digit (slots FB_NB + fb_xslot(m - 1, part)), in front of the reduction
__global__ void __launch_bounds__(64) k_fb_fold(uint8_t* __restrict__ buckets) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= (FB_XB >> FB_XPARTS_LOG)) return;
    G1XYZZ acc = xyzz_load(buckets + XYZZ_BYTES * (size_t)b);
#pragma unroll 1
    for (uint32_t j = 0; j < (1u << FB_XPARTS_LOG); ++j)
        acc = xyzz_add(acc, xyzz_load(buckets + XYZZ_BYTES * ((size_t)FB_NB + fb_xslot(b, j))));
    xyzz_store(buckets + XYZZ_BYTES * (size_t)b, acc);
}
__global__ void __launch_bounds__(64) k_fb_wsum(const uint8_t* __restrict__ vsum, const uint8_t* __restrict__ tsum,
                                                uint8_t* __restrict__ out) {
    __shared__ uint32_t lds[XYZZ_WORDS * 64];
    const uint32_t lane = threadIdx.x;
    G1XYZZ v = G1XYZZ::identity(), t = G1XYZZ::identity();
    if (lane < (uint32_t)FB_R2D_WINDOWS) {
        v = xyzz_load(vsum + XYZZ_BYTES * (size_t)lane);
        if (lane) t = xyzz_mul_small(xyzz_load(tsum + XYZZ_BYTES * (size_t)lane), lane);
    }
    const G1XYZZ sv = wave_sum_xyzz(v, lds);
    __syncthreads();
    const G1XYZZ st = wave_sum_xyzz(t, lds);
    if (lane == 0) {
        xyzz_store(out, sv);
        xyzz_store(out + XYZZ_BYTES, st);
    }
}

// ---- 4-lane cooperative doubling for the serial Horner tail --------------------------------------------
// A doubling is 9 field products on one lane (~2 300 VALU instructions, nothing to overlap with).  Its
// dependency graph is only three products deep:   {V = U^2, XX = X^2} -> {W = U*V, S = X*V, M^2, V*ZZ}
// -> {M*(S - X3), W*(4p - Y), W*ZZZ}.  Four lanes hold the point replicated, each computes one product per
// level (operands picked by lane id) and the results are broadcast back with cross-lane shuffles, so the
// chain is 3 products deep instead of 9.  Measured: k_msm_final 1.20 -> 0.97 ms at c = 16 (for a lone wave
// every instruction costs an issue slot, so the selects and shuffles are not free; a v_readlane variant that
// broadcasts through SGPRs was slower: 1.27 ms).
template <int SRC>
FP_INLINE Fq fq_bcast4(const Fq& v) {  // lane SRC of every group of 4 -> all 4: v_mov_b32 with a DPP quad_perm, no LDS
    Fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], SRC * 0x55, 0xF, 0xF, false);
    return r;
}
// Lane-dependent operand selection inside a group of 4, as straight-line v_cndmask_b32 with the lane pattern in an
// SGPR pair (bit i set: lane i takes `yes`).  Written as inline asm on purpose: the C ternaries this replaces were
// compiled into exec-masked branches per limb plus scratch traffic (about 15 instructions per limb instead of one
// to three), which made the cooperative doubling no faster than the one-lane formula (tools/ubench_latency.hip).
constexpr uint64_t QUAD_LANE0 = 0x1111111111111111ull, QUAD_LANE1 = QUAD_LANE0 << 1, QUAD_LANE2 = QUAD_LANE0 << 2,
                   QUAD_LANE3 = QUAD_LANE0 << 3;
FP_INLINE Fq fq_pick(uint64_t mask, const Fq& yes, const Fq& no) {
    Fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r.l[i]) : "v"(no.l[i]), "v"(yes.l[i]), "s"(mask));
    return r;
}
// 2 * p with p replicated in the 4 lanes of a group.  Same formulas and bounds as xyzz_double (dbl-2008-s-1),
// Y3 = M*(S - X3 + 6p) + W*(4p - Y) as a sum of two products [4].  Three rounds of one multiplication per lane:
//   round 1   lane 0: V = U^2      lane 1: XX = X^2
//   round 2   lane 0: W = U*V      lane 1: S = X*V      lane 2: M^2        lane 3: ZZ3 = V*ZZ
//   round 3   lane 0: M*(S-X3)     lane 1: W*(-Y)       lane 2,3: ZZZ3 = W*ZZZ
FP_INLINE G1XYZZ xyzz_double_par4(const G1XYZZ& p) {
    if (p.is_identity()) return p;  // uniform: the state is replicated
    const Fq u = FQ_DBL(p.y);                                           // [8]
    const Fq r1 = FQ_SQR(fq_pick(QUAD_LANE0, u, p.x));
    const Fq v = fq_bcast4<0>(r1), xx = fq_bcast4<1>(r1);
    const Fq m = fp_triple<FqParams>(xx);                               // [6]
    const Fq a2 = fq_pick(QUAD_LANE0, u, fq_pick(QUAD_LANE1, p.x, fq_pick(QUAD_LANE2, m, v)));
    const Fq b2 = fq_pick(QUAD_LANE2, m, fq_pick(QUAD_LANE3, p.zz, v));
    const Fq r2 = FQ_MUL(a2, b2);
    const Fq w = fq_bcast4<0>(r2), s = fq_bcast4<1>(r2), mm = fq_bcast4<2>(r2), zz3 = fq_bcast4<3>(r2);
    G1XYZZ o;
    o.x = fp_sub2<4, FqParams>(mm, s);                                  // [6]
    const Fq d = FQ_SUB(6, s, o.x);                                     // [8]
    const Fq ny = fp_neg<4, FqParams>(p.y);                             // [4]
    const Fq r3 = FQ_MUL(fq_pick(QUAD_LANE0, m, w), fq_pick(QUAD_LANE0, d, fq_pick(QUAD_LANE1, ny, p.zzz)));
    o.y = FQ_ADD(fq_bcast4<0>(r3), fq_bcast4<1>(r3));                   // [2] + [2] -> [4]
    o.zz = zz3;
    o.zzz = fq_bcast4<2>(r3);
    return o;
}

// a + b with both points replicated in the 4 lanes of a group: add-2008-s in four rounds of one multiplication per
// lane instead of 14 in a row (same exceptional cases as xyzz_add; they are uniform inside a group).
//   round 1   U1 = X1*ZZ2       U2 = X2*ZZ1       S1 = Y1*ZZZ2        S2 = Y2*ZZZ1
//   round 2   PP = P^2          RR = R^2          ZZ12 = ZZ1*ZZ2      ZZZ12 = ZZZ1*ZZZ2      (P = U2-U1, R = S2-S1)
//   round 3   PPP = P*PP        Q = U1*PP         ZZ3 = ZZ12*PP       -
//   round 4   R*(Q-X3)          (-S1)*PPP         ZZZ3 = ZZZ12*PPP    (same)                 (X3 = RR-PPP-2Q)
FP_INLINE G1XYZZ xyzz_add_par4(const G1XYZZ& a, const G1XYZZ& b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    const Fq a1 = fq_pick(QUAD_LANE0, a.x, fq_pick(QUAD_LANE1, b.x, fq_pick(QUAD_LANE2, a.y, b.y)));
    const Fq b1 = fq_pick(QUAD_LANE0, b.zz, fq_pick(QUAD_LANE1, a.zz, fq_pick(QUAD_LANE2, b.zzz, a.zzz)));
    const Fq r1 = FQ_MUL(a1, b1);                                       // 8*2 -> [2]
    const Fq u1 = fq_bcast4<0>(r1), u2 = fq_bcast4<1>(r1), s1 = fq_bcast4<2>(r1), s2 = fq_bcast4<3>(r1);
    const Fq p = FQ_SUB(2, u2, u1);                                     // [4]
    const Fq r = FQ_SUB(2, s2, s1);                                     // [4]
    if (fp_maybe_zero_mod<4, FqParams>(p)) {
        if (fp_is_zero_mod<4, FqParams>(p)) {
            if (fp_is_zero_mod<4, FqParams>(r)) return xyzz_double(a);
            return G1XYZZ::identity();
        }
    }
    const Fq a2 = fq_pick(QUAD_LANE0, p, fq_pick(QUAD_LANE1, r, fq_pick(QUAD_LANE2, a.zz, a.zzz)));
    const Fq b2 = fq_pick(QUAD_LANE2, b.zz, fq_pick(QUAD_LANE3, b.zzz, a2));
    const Fq r2 = FQ_MUL(a2, b2);                                       // 16 -> [2]
    const Fq pp = fq_bcast4<0>(r2), rr = fq_bcast4<1>(r2), zz12 = fq_bcast4<2>(r2), zzz12 = fq_bcast4<3>(r2);
    const Fq r3 = FQ_MUL(fq_pick(QUAD_LANE0, p, fq_pick(QUAD_LANE1, u1, zz12)), pp);   // [2]
    const Fq ppp = fq_bcast4<0>(r3), q = fq_bcast4<1>(r3);
    G1XYZZ o;
    o.zz = fq_bcast4<2>(r3);
    o.x = fp_sub_sub2<6, FqParams>(rr, ppp, q);                         // [8]
    const Fq a4 = fq_pick(QUAD_LANE0, r, fq_pick(QUAD_LANE1, fp_neg<2, FqParams>(s1), zzz12));
    const Fq b4 = fq_pick(QUAD_LANE0, FQ_SUB(8, q, o.x), ppp);          // [10] | [2]
    const Fq r4 = FQ_MUL(a4, b4);                                       // 4*10 -> [2]
    o.y = FQ_ADD(fq_bcast4<0>(r4), fq_bcast4<1>(r4));                   // [4]
    o.zzz = fq_bcast4<2>(r4);
    return o;
}

// ---- 4-lane cooperative versions of the latency-shaped tail kernels ------------------------------------------
// The bucket reduction and the window sums are chains of dependent additions (2*seg + a double-and-add per segment,
// then a strided sum and a tree per window): what bounds them on small and medium MSMs is the length of one chain,
// not the number of chains.  With 4 lanes per chain every addition is 4 products deep instead of 14.
__device__ __noinline__ G1XYZZ xyzz_mul_small_par4(const G1XYZZ& p, uint32_t k) {
    G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
    for (int bit = 31 - __clz((int)(k | 1u)); bit >= 0; --bit) {
        acc = xyzz_double_par4(acc);
        if ((k >> bit) & 1u) acc = xyzz_add_par4(acc, p);
    }
    return acc;
}
__global__ void __launch_bounds__(BLOCK, 2) k_msm_reduce_segments_par4(const uint8_t* __restrict__ buckets, uint32_t NB,
                                                                    uint32_t seg, uint32_t spw, uint32_t total,
                                                                    uint8_t* __restrict__ segsum) {
    const uint32_t t = (blockIdx.x * BLOCK + threadIdx.x) >> 2;   // one segment per group of 4 lanes
    if (t >= total) return;
    const uint32_t w = t / spw, sidx = t - w * spw;
    const uint32_t a = sidx * seg;
    const uint8_t* B = buckets + XYZZ_BYTES * ((size_t)w * NB + a);
    G1XYZZ running = G1XYZZ::identity(), acc = G1XYZZ::identity();
#pragma unroll 1
    for (int j = (int)seg - 1; j >= 0; --j) {
        running = xyzz_add_par4(running, xyzz_load(B + XYZZ_BYTES * (size_t)j));
        acc = xyzz_add_par4(acc, running);
    }
    if (a != 0 && !running.is_identity()) acc = xyzz_add_par4(acc, xyzz_mul_small_par4(running, a));
    if ((threadIdx.x & 3) == 0) xyzz_store(segsum + XYZZ_BYTES * (size_t)t, acc);
}
// k_msm_bucket_combine for few buckets (small MSMs with narrow windows: 4 096 buckets of 8 slice sums at 4 proofs): the seven
// one-lane additions in a row were 47 us at the head of an evaluation's tail.  Here a bucket has 2 LPB lanes — LPB / 2 groups of
// four, each adds two slice sums cooperatively, then the groups fold as a tree (the partner's sum comes over by lane shuffle):
// log2(LPB) cooperative additions deep.
FP_INLINE G1XYZZ xyzz_shfl_down(const G1XYZZ& v, int delta) {
    G1XYZZ o;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        o.x.l[i] = (uint32_t)__shfl_down((int)v.x.l[i], delta, 64);
        o.y.l[i] = (uint32_t)__shfl_down((int)v.y.l[i], delta, 64);
        o.zz.l[i] = (uint32_t)__shfl_down((int)v.zz.l[i], delta, 64);
        o.zzz.l[i] = (uint32_t)__shfl_down((int)v.zzz.l[i], delta, 64);
    }
    return o;
}
template <int LPB>
__global__ void __launch_bounds__(BLOCK) k_msm_bucket_combine_par4(const uint8_t* __restrict__ parts,
                                                                   const uint32_t* __restrict__ hist, uint32_t nbt,
                                                                   uint32_t big, uint8_t* __restrict__ buckets) {
    static_assert(LPB == 2 || LPB == 4 || LPB == 8, "lanes per bucket of the accumulation");
    constexpr int LANES = 2 * LPB;
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t key = t / LANES, l = t % LANES, q = l >> 2;
    if (key >= nbt) return;                                  // (whole lane groups: BLOCK is a multiple of LANES)
    const bool whole = hist[key] <= big;                     // an over-long bucket's sum is in slot 0 (chunk path)
    const uint8_t* P = parts + XYZZ_BYTES * (size_t)key * LPB;
    G1XYZZ a = (whole || q == 0) ? xyzz_load(P + XYZZ_BYTES * (size_t)(2 * q)) : G1XYZZ::identity();
    G1XYZZ acc = xyzz_add_par4(a, whole ? xyzz_load(P + XYZZ_BYTES * (size_t)(2 * q + 1)) : G1XYZZ::identity());
#pragma unroll
    for (int stride = 4; stride < LANES; stride <<= 1) {
        G1XYZZ o = xyzz_shfl_down(acc, stride);
        if (l % (2 * stride) >= 4) o = G1XYZZ::identity();   // not a receiver at this level (its partner may be another bucket's lane)
        acc = xyzz_add_par4(acc, o);
    }
    if (l == 0) xyzz_store(buckets + XYZZ_BYTES * (size_t)key, acc);
}

constexpr int PAR4_GROUPS = 128, PAR4_THREADS = 4 * PAR4_GROUPS;
// sum of one (replicated) point per group over a PAR4_THREADS workgroup; the result is replicated in group 0
__device__ __noinline__ G1XYZZ block_sum_xyzz_par4(G1XYZZ v, uint32_t* lds) {
    const int g = threadIdx.x >> 2, l = threadIdx.x & 3;
    if (l == 0) lds_put_xyzz(lds, g, v);
    __syncthreads();
#pragma unroll 1
    for (int s = PAR4_GROUPS / 2; s >= 1; s >>= 1) {
        if (g < s) {
            v = xyzz_add_par4(v, lds_get_xyzz(lds, g + s));
            if (l == 0) lds_put_xyzz(lds, g, v);
        }
        __syncthreads();
    }
    return v;
}
__global__ void __launch_bounds__(PAR4_THREADS) k_msm_window_sum_par4(const uint8_t* __restrict__ segsum, uint32_t spw,
                                                                      uint8_t* __restrict__ wsum) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    const uint32_t w = blockIdx.x;
    G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
    for (uint32_t s = threadIdx.x >> 2; s < spw; s += PAR4_GROUPS)
        acc = xyzz_add_par4(acc, xyzz_load(segsum + XYZZ_BYTES * ((size_t)w * spw + s)));
    G1XYZZ tot = block_sum_xyzz_par4(acc, lds);
    if (threadIdx.x == 0) xyzz_store(wsum + XYZZ_BYTES * (size_t)w, tot);
}

// result = sum_w 2^(c*w) * wsum[w]  (Horner, top window first).  Writes the XYZZ value (Montgomery, for
// on-device consumers such as the eval tail) and the canonical Jacobian encoding of the C ABI.
// One wave; the point is replicated across lanes, lanes cooperate in groups of 4 on every doubling.
__global__ void __launch_bounds__(64) k_msm_final(const uint8_t* __restrict__ wsum, int c, int W,
                                                  uint8_t* __restrict__ out_xyzz, uint8_t* __restrict__ out_jac) {
    // one workgroup (one wave) per MSM of a batch: its W windows start at wsum[blockIdx.x * W]
    wsum += XYZZ_BYTES * (size_t)blockIdx.x * W;
    if (out_xyzz) out_xyzz += XYZZ_BYTES * (size_t)blockIdx.x;
    if (out_jac) out_jac += 96 * (size_t)blockIdx.x;
    // this wave carries the whole latency chain: let it win issue arbitration against the bulk kernels of the
    // next MSM that share its SIMD in overlap mode
    __builtin_amdgcn_s_setprio(3);
    G1XYZZ acc = xyzz_load(wsum + XYZZ_BYTES * (size_t)(W - 1));
#pragma unroll 1
    for (int w = W - 2; w >= 0; --w) {
#pragma unroll 1
        for (int k = 0; k < c; ++k) acc = xyzz_double_par4(acc);
        acc = xyzz_add_par4(acc, xyzz_load(wsum + XYZZ_BYTES * (size_t)w));
    }
    if (threadIdx.x == 0) {
        if (out_xyzz) xyzz_store(out_xyzz, acc);
        if (out_jac) jac_store_canonical(out_jac, jac_from_xyzz(acc));
    }
}

// eval()'s tail (evaluation.rs:198-200): acc = msm_result + sum of the scalar-less points (canonical affine)
__global__ void __launch_bounds__(BLOCK) k_eval_tail(const uint8_t* __restrict__ msm_xyzz,
                                                     const uint8_t* __restrict__ pts_aff, size_t n,
                                                     uint8_t* __restrict__ out_jac, uint32_t* flags) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    G1XYZZ acc = G1XYZZ::identity();
    if (threadIdx.x == 0 && msm_xyzz) acc = xyzz_load(msm_xyzz);
    uint32_t bad = 0;
    for (size_t i = threadIdx.x; i < n; i += BLOCK) xyzz_add_affine(acc, affine_load_canonical(pts_aff + 64 * i, bad));
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    G1XYZZ tot = block_sum_xyzz(acc, lds);
    if (threadIdx.x == 0) jac_store_canonical(out_jac, jac_from_xyzz(tot));
}

// both sides of evaluate_multiopen_proof in one launch (one workgroup each), straight to `to_value`:
// out_aff[64 * side] = to_affine(msm_result + sum of the scalar-less points)   (evaluation.rs:198-200, verify.rs:730-731)
__global__ void __launch_bounds__(BLOCK) k_eval_tail_affine2(const uint8_t* __restrict__ xyzz0,
                                                             const uint8_t* __restrict__ xyzz1,
                                                             const uint8_t* __restrict__ pts0,
                                                             const uint8_t* __restrict__ pts1, size_t n0, size_t n1,
                                                             uint8_t* __restrict__ out_aff, uint32_t* flags,
                                                             uint8_t* __restrict__ out_xyzz = nullptr) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    const bool side = blockIdx.x != 0;
    const uint8_t* msm_xyzz = side ? xyzz1 : xyzz0;
    const uint8_t* pts_aff = side ? pts1 : pts0;
    const size_t n = side ? n1 : n0;
    G1XYZZ acc = G1XYZZ::identity();
    if (threadIdx.x == 0 && msm_xyzz) acc = xyzz_load(msm_xyzz);
    uint32_t bad = 0;
    G1XYZZ tot;
    if (n <= 8) {
        // the usual case (a handful of scalar-less points, often none): one lane adds them in a row — the workgroup tree
        // below is eight levels of general additions (~50 us at a lone wave's latency) whatever the count
        if (threadIdx.x == 0)
            for (size_t i = 0; i < n; ++i) xyzz_add_affine(acc, affine_load_canonical(pts_aff + 64 * i, bad));
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        tot = acc;
    } else {
        for (size_t i = threadIdx.x; i < n; i += BLOCK) xyzz_add_affine(acc, affine_load_canonical(pts_aff + 64 * i, bad));
        if (bad) atomicOr(flags, FLAG_NONCANONICAL);
        tot = block_sum_xyzz(acc, lds);
    }
    if (threadIdx.x == 0) {
        if (out_xyzz) {
            // the coordinates as they are (X, Y, ZZ, ZZZ; canonical integers): the caller divides on the host, where one field
            // inversion is ~10 us — here it is ~55 us of a lone wave's latency at the very end of the evaluation's chain
            uint8_t* o = out_xyzz + 128 * blockIdx.x;
            fp_store<FqParams>(o, fp_from_mont<FqParams>(tot.x));
            fp_store<FqParams>(o + 32, fp_from_mont<FqParams>(tot.y));
            fp_store<FqParams>(o + 64, fp_from_mont<FqParams>(tot.zz));
            fp_store<FqParams>(o + 96, fp_from_mont<FqParams>(tot.zzz));
        } else {
            const G1Affine a = affine_from_xyzz(tot);
            fp_store<FqParams>(out_aff + 64 * blockIdx.x, fp_from_mont<FqParams>(a.x));
            fp_store<FqParams>(out_aff + 64 * blockIdx.x + 32, fp_from_mont<FqParams>(a.y));
        }
    }
}

}  // namespace h2agg
