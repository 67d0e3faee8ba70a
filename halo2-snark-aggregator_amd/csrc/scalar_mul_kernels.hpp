// Windowed scalar multiplication for gfx950: the batched form of MockEccChip::scalar_mul / scalar_mul_constant
//   halo2-snark-aggregator-api/src/mock/arith/ecc.rs:88-104      (`rhs * lhs`: a 254-step double-and-add per call)
// and of the loop body of assign_instance_commitment (verify.rs:623-635).  The window shape is the reference's own
// "windowed point-add/double" (halo2-ecc-circuit-lib/src/chips/ecc_chip.rs:70,86-138: CONFIG_WINDOW_SIZE = 4, a table of
// multiples, 4 doublings + 1 addition per window); the group element produced is the same.
//
// Round 1 ran a bit-serial ladder, one lane per point: 256 doublings + a divergent conditional addition per bit, ~1.9 ms
// of pure dependent latency for any n (the call is latency-bound: a lone wave issues one instruction per ~4.3 cycles).
// Here the chain is cut three ways:
//   * GLV: k = k1 + lambda*k2 with |k_i| < 2^127 and phi(x, y) = (beta*x, y) — one shared doubling chain of 128 instead
//     of 256; phi(T) of a table entry is a single multiplication of its X by beta;
//   * signed window-4 digits in [-8, 8]: 33 digit positions, one table of 1P .. 8P per point (in LDS, XYZZ), at most two
//     additions per position (one per GLV half), no per-bit branches;
//   * four lanes per point (xyzz_double_par4 / xyzz_add_par4 of msm_kernels.hpp): a doubling is 3 products deep instead
//     of 9, an addition 4 instead of 14.
// Chain per point: 7 table additions + 32 x (4 doublings + <= 2 additions) ~ 0.5 ms, against 1.9 ms.
//
// k_bases_generate (workload generation: P_i = k_i * G) uses a fixed-base comb instead: a table d * 2^(8w) * G,
// d = 1..255, w = 0..31 (8160 affine points, 510 KiB, built once per context) turns k * G into <= 32 mixed additions and
// no doubling at all.
#pragma once
#include "msm_kernels.hpp"

namespace h2agg {

constexpr int SM_THREADS = 128;              // 32 points per workgroup, 4 lanes each
constexpr int SM_GROUPS = SM_THREADS / 4;
constexpr int SM_TABLE = 8;                  // 1P .. 8P

// signed base-16 digits of a 127-bit magnitude (four 32-bit words, bit 127 clear): 33 digits in [-8, 8],
// m = sum_j d_j 16^j.  Returned packed: digit j = ((lo >> (4*(j%16))) & 15) interpreted through `neg` bit j.
struct W4Digits {
    uint32_t mag[5];   // 4 bits per digit magnitude (0..8): 33 digits -> 132 bits
    uint64_t neg;      // bit j set: digit j is negative (bits 0..32)
};
FP_INLINE W4Digits w4_recode(const uint32_t (&m)[4]) {
    W4Digits r;
#pragma unroll
    for (int i = 0; i < 5; ++i) r.mag[i] = 0;
    r.neg = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 33; ++j) {
        const uint32_t nib = (j < 32 ? ((m[j >> 3] >> (4 * (j & 7))) & 15u) : 0u) + carry;   // 0..16
        const bool ng = nib > 8u;
        carry = ng ? 1u : 0u;
        const uint32_t mg = ng ? 16u - nib : nib;                                           // 0..8
        r.mag[j >> 3] |= mg << (4 * (j & 7));
        if (ng) r.neg |= (uint64_t)1 << j;
    }
    return r;
}
FP_INLINE uint32_t w4_mag(const W4Digits& d, int j) { return (d.mag[j >> 3] >> (4 * (j & 7))) & 15u; }

FP_INLINE Fq fq_beta() {
    Fq b;
#pragma unroll
    for (int i = 0; i < NL; ++i) b.l[i] = GlvConst::BETA_MONT[i];
    return b;
}

// LDS table layout: word k of entry e of group g at tab[(e * XYZZ_WORDS + k) * SM_GROUPS + g] (conflict-free across groups)
FP_INLINE void sm_tab_put(uint32_t* tab, int g, int e, const G1XYZZ& p) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        tab[((e * XYZZ_WORDS) + k) * SM_GROUPS + g] = p.x.l[k];
        tab[((e * XYZZ_WORDS) + NL + k) * SM_GROUPS + g] = p.y.l[k];
        tab[((e * XYZZ_WORDS) + 2 * NL + k) * SM_GROUPS + g] = p.zz.l[k];
        tab[((e * XYZZ_WORDS) + 3 * NL + k) * SM_GROUPS + g] = p.zzz.l[k];
    }
}
FP_INLINE G1XYZZ sm_tab_get(const uint32_t* tab, int g, int e) {
    G1XYZZ p;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        p.x.l[k] = tab[((e * XYZZ_WORDS) + k) * SM_GROUPS + g];
        p.y.l[k] = tab[((e * XYZZ_WORDS) + NL + k) * SM_GROUPS + g];
        p.zz.l[k] = tab[((e * XYZZ_WORDS) + 2 * NL + k) * SM_GROUPS + g];
        p.zzz.l[k] = tab[((e * XYZZ_WORDS) + 3 * NL + k) * SM_GROUPS + g];
    }
    return p;
}

// s * P for the (replicated) point of this group of four lanes; `tab` is the workgroup's table area
FP_INLINE G1XYZZ g1_scalar_mul_w4_par4(const G1Affine& base, const U256& s, uint32_t* tab, uint32_t* flags) {
    const int g = threadIdx.x >> 2;
    U256 d;
    if (!glv_decompose(s, d)) atomicOr(flags, FLAG_NONCANONICAL);   // cannot happen for s < r; the bound is still checked
    const uint32_t m1[4] = {d.w[0], d.w[1], d.w[2], d.w[3] & 0x7fffffffu};
    const uint32_t m2[4] = {d.w[4], d.w[5], d.w[6], d.w[7] & 0x7fffffffu};
    const bool sg1 = (d.w[3] >> 31) != 0, sg2 = (d.w[7] >> 31) != 0;
    const W4Digits d1 = w4_recode(m1), d2 = w4_recode(m2);
    // table 1P .. 8P: 2P by doubling, then +P each
    G1XYZZ t = G1XYZZ::from_affine(base);
    const G1XYZZ p1 = t;
    if ((threadIdx.x & 3) == 0) sm_tab_put(tab, g, 0, t);
    t = xyzz_double_par4(t);
    if ((threadIdx.x & 3) == 0) sm_tab_put(tab, g, 1, t);
#pragma unroll 1
    for (int e = 2; e < SM_TABLE; ++e) {
        t = xyzz_add_par4(t, p1);
        if ((threadIdx.x & 3) == 0) sm_tab_put(tab, g, e, t);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();     // the four lanes of a group are in one wave: the table is visible to them
    const Fq beta = fq_beta();
    G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
    for (int j = 32; j >= 0; --j) {
        if (j != 32) {
#pragma unroll 1
            for (int k = 0; k < 4; ++k) acc = xyzz_double_par4(acc);
        }
        const uint32_t a1 = w4_mag(d1, j), a2 = w4_mag(d2, j);
        if (a1) {
            G1XYZZ q = sm_tab_get(tab, g, (int)a1 - 1);
            if ((((d1.neg >> j) & 1u) != 0) != sg1) q.y = fp_neg<4, FqParams>(q.y);
            acc = xyzz_add_par4(acc, q);
        }
        if (a2) {
            G1XYZZ q = sm_tab_get(tab, g, (int)a2 - 1);
            q.x = FQ_MUL(q.x, beta);                                  // phi: X -> beta * X  (8 * 1 / 169 + 1 -> [2])
            if ((((d2.neg >> j) & 1u) != 0) != sg2) q.y = fp_neg<4, FqParams>(q.y);
            acc = xyzz_add_par4(acc, q);
        }
    }
    return acc;
}

// MockEccChip::scalar_mul / scalar_mul_constant over n (affine base, scalar) pairs  (mock/arith/ecc.rs:88-104)
__global__ void __launch_bounds__(SM_THREADS) k_g1_batch_scalar_mul_w4(const uint8_t* __restrict__ bases,
                                                                        const uint8_t* __restrict__ scalars, size_t n,
                                                                        uint8_t* __restrict__ out, uint32_t* flags) {
    __shared__ uint32_t tab[SM_TABLE * XYZZ_WORDS * SM_GROUPS];
    const size_t ngroups_total = (size_t)gridDim.x * SM_GROUPS;
    // every group of a workgroup walks the same number of rounds so that the wave-level barrier inside stays uniform
    for (size_t i0 = (size_t)blockIdx.x * SM_GROUPS; i0 < n; i0 += ngroups_total) {
        const size_t i = i0 + (threadIdx.x >> 2);
        const bool live = i < n;
        const size_t ii = live ? i : n - 1;
        uint32_t bad = 0;
        const G1Affine p = affine_load_canonical(bases + 64 * ii, bad);
        const U256 s = u256_load(scalars + 32 * ii);
        bad |= !u256_is_canonical_fr(s);
        if (bad && live) atomicOr(flags, FLAG_NONCANONICAL);
        G1XYZZ r = G1XYZZ::identity();
        if (!p.is_identity()) r = g1_scalar_mul_w4_par4(p, s, tab, flags);
        if (live && (threadIdx.x & 3) == 0) jac_store_canonical(out + 96 * i, jac_from_xyzz(r));
        __syncthreads();   // the table area is reused by the next round
    }
}

// ---- fixed-base comb for the generator ----------------------------------------------------------------------------
constexpr int COMB_BITS = 8, COMB_WINDOWS = 32, COMB_ROW = (1 << COMB_BITS) - 1;   // 32 x 255 affine points

// table[w * 255 + (d - 1)] = d * 2^(8w) * G (Montgomery affine, 64 B); one group of four lanes per entry.
// bases != nullptr: the same rows for each of nb resident bases instead of the generator, table[(b * 32 + w) * 255 + (d - 1)] =
// d * 2^(8w) * B_b — the comb of a FIXED table's leading bases (k_comb_msm below).
__global__ void __launch_bounds__(SM_THREADS) k_comb_table_build(uint8_t* __restrict__ table, uint32_t* flags,
                                                                 const uint8_t* __restrict__ bases = nullptr, uint32_t nb = 1) {
    __shared__ uint32_t tab[SM_TABLE * XYZZ_WORDS * SM_GROUPS];
    const size_t per_base = (size_t)COMB_WINDOWS * COMB_ROW;
    const size_t total = per_base * nb;
    const size_t ngroups_total = (size_t)gridDim.x * SM_GROUPS;
    for (size_t e0 = (size_t)blockIdx.x * SM_GROUPS; e0 < total; e0 += ngroups_total) {
        const size_t e = e0 + (threadIdx.x >> 2);
        const bool live = e < total;
        const size_t ee = live ? e : total - 1;
        const size_t bi = ee / per_base, er = ee - bi * per_base;
        const uint32_t w = (uint32_t)(er / COMB_ROW), dgt = (uint32_t)(er % COMB_ROW) + 1u;
        U256 s;
#pragma unroll
        for (int k = 0; k < 8; ++k) s.w[k] = 0;
        s.w[w >> 2] = dgt << (8 * (w & 3));                             // d * 2^(8w) < 2^256; may exceed r: reduce below
        // d * 2^(8w) can be >= r only in the top window (w = 31, d >= 0x31): subtract r until canonical (at most 5 times)
#pragma unroll 1
        for (int it = 0; it < 6; ++it) {
            if (u256_is_canonical_fr(s)) break;
            uint64_t br = 0;
            const uint32_t rw[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint64_t dd = (uint64_t)s.w[k] - rw[k] - br;
                s.w[k] = (uint32_t)dd;
                br = (dd >> 32) & 1;
            }
        }
        G1Affine gen;
        if (bases) {
            gen = affine_load(bases + 64 * bi);
        } else {
            gen.x = Fq::one();
            gen.y = FQ_DBL(Fq::one());
        }
        G1XYZZ r = G1XYZZ::identity();
        if (!gen.is_identity()) r = g1_scalar_mul_w4_par4(gen, s, tab, flags);
        if (live && (threadIdx.x & 3) == 0) affine_store(table + 64 * e, affine_from_xyzz(r));
        __syncthreads();
    }
}

// bases[i] = k_i * G through the comb: sum over the 32 bytes of k_i of table[w][byte - 1] (mixed additions only)
__global__ void __launch_bounds__(BLOCK) k_bases_generate_comb(const uint8_t* __restrict__ k, size_t n,
                                                               const uint8_t* __restrict__ table,
                                                               uint8_t* __restrict__ out, uint32_t* flags) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const U256 s = u256_load(k + 32 * i);
        if (!u256_is_canonical_fr(s)) atomicOr(flags, FLAG_NONCANONICAL);
        G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
        for (int w = 0; w < COMB_WINDOWS; ++w) {
            const uint32_t d = (s.w[w >> 2] >> (8 * (w & 3))) & 0xffu;
            if (d) xyzz_add_affine(acc, affine_load(table + 64 * ((size_t)w * COMB_ROW + d - 1)));
        }
        affine_store(out + 64 * i, affine_from_xyzz(acc));
    }
}

// Small multi_exps over the LEADING bases of a fixed table (an instance column of a few dozen public inputs against
// params.g_lagrange: assign_instance_commitment, verify.rs:574-649): sum_i v_i B_i = sum over (i, byte position w) of
// table[i][w][byte_w(v_i)] — no doublings, no buckets, no serial tail: every thread adds its share of the n x 32 table entries
// with mixed additions, one tree per workgroup.  One workgroup per MSM of the batch (scalars [batch][n]); canonical Jacobian out.
constexpr int COMB_MSM_MAX = 256;   // bases per table that get a comb (133 MB); longer MSMs take the bucket path
__global__ void __launch_bounds__(BLOCK) k_comb_msm(const uint8_t* __restrict__ table, const uint8_t* __restrict__ scalars,
                                                    uint32_t n, uint8_t* __restrict__ out_jac, uint32_t* flags) {
    __shared__ uint32_t lds[XYZZ_WORDS * BLOCK];
    const uint8_t* sc = scalars + 32 * (size_t)n * blockIdx.x;
    G1XYZZ acc = G1XYZZ::identity();
    bool bad = false;
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < n * (uint32_t)COMB_WINDOWS; e += BLOCK) {
        const uint32_t i = e / COMB_WINDOWS, w = e % COMB_WINDOWS;
        if (w == 0) bad |= !u256_is_canonical_fr(u256_load(sc + 32 * (size_t)i));
        const uint32_t d = sc[32 * (size_t)i + w];
        if (d) xyzz_add_affine(acc, affine_load(table + 64 * (((size_t)i * COMB_WINDOWS + w) * COMB_ROW + d - 1)));
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    const G1XYZZ tot = block_sum_xyzz(acc, lds);
    if (threadIdx.x == 0) jac_store_canonical(out_jac + 96 * (size_t)blockIdx.x, jac_from_xyzz(tot));
}

}  // namespace h2agg
