// Limb-parallel field and group arithmetic for the LATENCY-shaped tails of the MSM (gfx950).
//
// Stands behind the same halo2curves operators as fp.hpp / g1.hpp (the Horner chain of a multi_exp: mock/arith/ecc.rs:106-129 ->
// c doublings + one addition per window).  A lone wave is bound by the number of DEPENDENT instructions, not by throughput: a
// Montgomery product on one lane is a chain of ~235 of them (162 multiply-adds).  Here a field element lives ACROSS lanes — limb j
// in lane j of a DPP row of 16 (lanes 9..15 hold zeros) — and the product is operand scanning with one multiply-add per lane
// per step: broadcast a_i (row_share), acc += a_i * b_j, quotient digit from lane 0 (row_share), acc += m * p_j, shift the
// accumulators one lane down (row_shl) and fold lane 0's carry in: 12 dependent instructions per step, 9 steps, plus a carry
// pass: ~120 instead of ~235.  A wavefront's four rows run FOUR independent products at once, which is exactly the width of
// the group law (a doubling is three rounds of <= 4 products, a general addition four): one point per wave, its coordinates
// replicated in all rows, each row computes one product per round and the results are handed to all rows with CDNA4's
// v_permlane32_swap / v_permlane16_swap.
//
// Limbs are "nearly tight": 29 bits plus at most a few units (one carry pass, never a ripple).  A subtraction a - b adds K p
// in a BORROWED form (every limb of the constant below the top is >= 2^29, the top limb one less: the same integer), so no limb
// difference is ever negative: no signed limbs, no borrow chains.  It needs value(b) <= (K - 1) p (the top limb of the
// constant must not fall below b's).  Bounds follow g1.hpp's invariants (X < 8, Y < 4, ZZ, ZZZ < 2, in multiples of p) with
// K one more than there (X < 9 here: the sum PPP + 2 Q [6] is subtracted with 7 p).
#pragma once
#include "g1.hpp"

namespace h2agg {

struct LpConst {
    uint32_t pj;       // limb j of p (0 in lanes 9..15)
    uint32_t c3, c5, c7, c9, c11;   // limb j of 3 p / 5 p / 7 p / 9 p / 11 p in borrowed form
    uint32_t maskj;    // 2^29 - 1 below the top limb, all ones in it (and 0 beyond)
    uint32_t shj;      // carry shift: 29 below the top limb, 31 from it on (nothing leaves the top limb)
    uint32_t lane0;    // all ones in lane 0 of every row
    int j, row;
};
FP_INLINE LpConst lp_const() {
    LpConst k;
    const int lane = threadIdx.x & 63;
    k.j = lane & 15;
    k.row = lane >> 4;
    auto borrowed = [&](int K) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t kp = km_limb<FqParams>(K, i);
            // limb 0 gets +2^29, limbs 1..7 +2^29 - 1, limb 8 -1 (the same integer)
            const uint32_t ci = i == 0 ? kp + (1u << 29) : (i < 8 ? kp + (1u << 29) - 1u : kp - 1u);
            if (k.j == i) c = ci;
        }
        return c;
    };
    uint32_t p = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (k.j == i) p = FqParams::MOD[i];
    k.pj = p;
    k.c3 = borrowed(3);
    k.c5 = borrowed(5);
    k.c7 = borrowed(7);
    k.c9 = borrowed(9);
    k.c11 = borrowed(11);
    k.maskj = k.j < 8 ? M29 : (k.j == 8 ? 0xffffffffu : 0u);
    k.shj = k.j < 8 ? 29u : 31u;
    k.lane0 = k.j == 0 ? 0xffffffffu : 0u;
    return k;
}

template <int I>
FP_INLINE uint32_t lp_share(uint32_t v) {   // lane I of every row -> the whole row
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + I, 0xF, 0xF, false);
}
FP_INLINE uint32_t lp_from_above(uint32_t v) {   // lane j gets lane j + 1 of its row (the row's last lane gets 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true);
}
FP_INLINE uint32_t lp_from_below(uint32_t v) {   // lane j gets lane j - 1 of its row (lane 0 gets 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
}
// one carry pass over 32-bit limb values < 2^31: limbs come out < 2^29 + 4 (top limb: whatever the value needs)
FP_INLINE uint32_t lp_carry(uint32_t x, const LpConst& k) {
    const uint32_t c = x >> k.shj;
    return (x & k.maskj) + lp_from_below(c);
}
FP_INLINE uint32_t lp_add(uint32_t a, uint32_t b, const LpConst& k) { return lp_carry(a + b, k); }
template <int K>
FP_INLINE uint32_t lp_ck(const LpConst& k) {
    static_assert(K == 3 || K == 5 || K == 7 || K == 9 || K == 11, "borrowed constants exist for 3 p, 5 p, 7 p, 9 p, 11 p");
    return K == 3 ? k.c3 : (K == 5 ? k.c5 : (K == 7 ? k.c7 : (K == 9 ? k.c9 : k.c11)));
}
// a - b + K p.  REQUIRES value(b) <= (K - 1) p.  bound: A + K.
template <int K>
FP_INLINE uint32_t lp_sub(uint32_t a, uint32_t b, const LpConst& k) { return lp_carry(a + lp_ck<K>(k) - b, k); }
// K p - b.  REQUIRES value(b) <= (K - 1) p.  bound: K.
template <int K>
FP_INLINE uint32_t lp_neg(uint32_t b, const LpConst& k) { return lp_carry(lp_ck<K>(k) - b, k); }
FP_INLINE uint32_t lp_triple(uint32_t a, const LpConst& k) { return lp_carry(a * 3u, k); }

template <int I>
FP_INLINE void lp_mul_step(uint64_t& acc, uint32_t a, uint32_t b, const LpConst& k) {
    const uint32_t ai = lp_share<I>(a);
    acc += (uint64_t)ai * b;
    const uint32_t t0 = lp_share<0>((uint32_t)acc);
    const uint32_t m = (t0 * FqParams::NINV) & M29;
    acc += (uint64_t)m * k.pj;
    // one lane down; lane 0's (now divisible by 2^29) value leaves as a carry into the new lane 0
    const uint64_t c = acc >> 29;
    const uint32_t lo = lp_from_above((uint32_t)acc), hi = lp_from_above((uint32_t)(acc >> 32));
    const uint64_t cm = ((uint64_t)((uint32_t)(c >> 32) & k.lane0) << 32) | ((uint32_t)c & k.lane0);
    acc = (((uint64_t)hi << 32) | lo) + cm;
}
// per row: a * b / 2^261 mod p.  Bounds as fp_mul: inputs A p, B p -> (A B / 169 + 1) p.
FP_INLINE uint32_t lp_mul(uint32_t a, uint32_t b, const LpConst& k) {
    uint64_t acc = 0;
    lp_mul_step<0>(acc, a, b, k);
    lp_mul_step<1>(acc, a, b, k);
    lp_mul_step<2>(acc, a, b, k);
    lp_mul_step<3>(acc, a, b, k);
    lp_mul_step<4>(acc, a, b, k);
    lp_mul_step<5>(acc, a, b, k);
    lp_mul_step<6>(acc, a, b, k);
    lp_mul_step<7>(acc, a, b, k);
    lp_mul_step<8>(acc, a, b, k);
    // columns < 2^63 -> nearly tight limbs: the carry of a column is < 2^34 = 29 bits for the next lane + 5 for the one after
    const uint64_t c = acc >> 29;
    const uint32_t c_lo = (uint32_t)c & M29, c_hi = (uint32_t)(c >> 29);
    const uint32_t keep = k.j < 8 ? ((uint32_t)acc & M29) : (uint32_t)acc;   // (the value is < 2^261: the top column is < 2^29 by itself)
    uint32_t r = keep + lp_from_below(k.j < 8 ? c_lo : 0u);
    r += lp_from_below(lp_from_below(k.j < 7 ? c_hi : 0u));
    // (c_hi of lane 7 would belong two lanes up, beyond the top limb: it is zero because the value fits; lane 7's c_lo reaches lane 8)
    return lp_carry(r, k);
}

// ---- a point, one per wave: coordinate limbs replicated in the four rows
struct LpPoint {
    uint32_t x, y, zz, zzz;
};
FP_INLINE bool lp_is_identity(const LpPoint& p) {   // zz is the integer 0 (wave-uniform answer)
    return __builtin_amdgcn_ballot_w64(p.zz != 0) == 0;
}
FP_INLINE LpPoint lp_load(const void* rec, const LpConst& k) {   // an XYZZ record (144 B): lane j takes limb j of every coordinate
    const uint32_t* w = reinterpret_cast<const uint32_t*>(rec);
    LpPoint p;
    const int j = k.j < NL ? k.j : 0;
    p.x = k.j < NL ? w[j] : 0u;
    p.y = k.j < NL ? w[NL + j] : 0u;
    p.zz = k.j < NL ? w[2 * NL + j] : 0u;
    p.zzz = k.j < NL ? w[3 * NL + j] : 0u;
    return p;
}
// the four rows' values of v (the result of a round) at every row: r[q] = row q's v
FP_INLINE void lp_all_rows(uint32_t v, uint32_t (&r)[4]) {
    // v_permlane32_swap x, y: x's upper 32 lanes <-> y's lower 32 lanes.  With x = y = v: x = (rows 0, 1, 0, 1), y = (rows 2, 3, 2, 3)
    auto s32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    // v_permlane16_swap x, y: x's odd rows <-> y's even rows.  With x = y = (A, B, A, B): x = (A, A, A, A), y = (B, B, B, B)
    auto lo = __builtin_amdgcn_permlane16_swap(s32[0], s32[0], false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(s32[1], s32[1], false, false);
    r[0] = lo[0];
    r[1] = lo[1];
    r[2] = hi[0];
    r[3] = hi[1];
}
FP_INLINE uint32_t lp_pick(int row, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    return row == 0 ? v0 : (row == 1 ? v1 : (row == 2 ? v2 : v3));
}

// 2 p (dbl-2008-s-1, a = 0), three rounds of one product per row:
//   round 1   row 0: V = U^2      row 1: XX = X^2                                           (U = 2 Y)
//   round 2   row 0: W = U V      row 1: S = X V      row 2: MM = M^2    row 3: ZZ3 = V ZZ   (M = 3 XX)
//   round 3   row 0: M (S - X3)   row 1: W (-Y)       row 2: ZZZ3 = W ZZZ                    (X3 = MM - 2 S; Y3 = the sum of rows 0, 1)
FP_INLINE LpPoint lp_double(const LpPoint& p, const LpConst& k) {
    if (lp_is_identity(p)) return p;
    const uint32_t u = lp_add(p.y, p.y, k);                             // [8]
    uint32_t r[4];
    lp_all_rows(lp_mul(lp_pick(k.row, u, p.x, u, u), lp_pick(k.row, u, p.x, u, u), k), r);
    const uint32_t v = r[0], xx = r[1];
    const uint32_t m = lp_triple(xx, k);                                // XX [2] (49 / 169 + 1) -> [6]
    lp_all_rows(lp_mul(lp_pick(k.row, u, p.x, m, v), lp_pick(k.row, v, v, m, p.zz), k), r);
    const uint32_t w = r[0], s = r[1], mm = r[2], zz3 = r[3];
    LpPoint o;
    o.x = lp_sub<5>(mm, lp_add(s, s, k), k);                            // MM [2] - 2 S [4] + 5 p -> [7]
    const uint32_t d = lp_sub<9>(s, o.x, k), ny = lp_neg<5>(p.y, k);     // [11], [5]: 6 * 11 and 2 * 5 stay under 169
    lp_all_rows(lp_mul(lp_pick(k.row, m, w, w, w), lp_pick(k.row, d, ny, p.zzz, p.zzz), k), r);
    o.y = lp_add(r[0], r[1], k);
    o.zz = zz3;
    o.zzz = r[2];
    return o;
}

// ---- between the lane-parallel form and g1.hpp's one-lane form (through 4 x 64 words of LDS; `sm` is THIS WAVE's own scratch)
// No workgroup barrier: the LDS operations of one wave execute in order, the wave-scope fence keeps the compiler from moving
// them — so these work inside workgroups of several waves and inside wave-uniform branches.
FP_INLINE void lp_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
FP_INLINE Fq lp_collect(const uint32_t* sm) {   // row 0's limbs -> tight limbs on the calling lane
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)sm[i];
    return fp_normalize<FqParams>(x);
}
FP_INLINE G1XYZZ lp_to_single(const LpPoint& p, uint32_t* sm) {   // valid on every lane (all read row 0)
    const int lane = threadIdx.x & 63;
    sm[lane] = p.x;
    sm[64 + lane] = p.y;
    sm[128 + lane] = p.zz;
    sm[192 + lane] = p.zzz;
    lp_wave_fence();
    G1XYZZ g;
    g.x = lp_collect(sm);
    g.y = lp_collect(sm + 64);
    g.zz = lp_collect(sm + 128);
    g.zzz = lp_collect(sm + 192);
    lp_wave_fence();
    return g;
}
FP_INLINE LpPoint lp_from_single(const G1XYZZ& g, uint32_t* sm, const LpConst& k) {   // g: lane 0's value counts
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            sm[i] = g.x.l[i];
            sm[64 + i] = g.y.l[i];
            sm[128 + i] = g.zz.l[i];
            sm[192 + i] = g.zzz.l[i];
        }
    }
    lp_wave_fence();
    LpPoint p;
    const int j = k.j < NL ? k.j : 0;
    p.x = k.j < NL ? sm[j] : 0u;
    p.y = k.j < NL ? sm[64 + j] : 0u;
    p.zz = k.j < NL ? sm[128 + j] : 0u;
    p.zzz = k.j < NL ? sm[192 + j] : 0u;
    lp_wave_fence();
    return p;
}

// a + b (add-2008-s), four rounds of one product per row; the exceptional cases (an identity operand: decided here, wave-uniform;
// P = U2 - U1 = 0 mod p, i.e. b = +-a: a one-limb filter that cannot miss a multiple of p, then g1.hpp's exact xyzz_add on one lane)
//   round 1   U1 = X1 ZZ2        U2 = X2 ZZ1        S1 = Y1 ZZZ2          S2 = Y2 ZZZ1
//   round 2   PP = P^2           RR = R^2           ZZ12 = ZZ1 ZZ2        ZZZ12 = ZZZ1 ZZZ2        (P = U2 - U1, R = S2 - S1)
//   round 3   PPP = P PP         Q = U1 PP          ZZ3 = ZZ12 PP
//   round 4   R (Q - X3)         (-S1) PPP          ZZZ3 = ZZZ12 PPP                               (X3 = RR - PPP - 2 Q)
FP_INLINE LpPoint lp_add_points(const LpPoint& a, const LpPoint& b, const LpConst& k, uint32_t* sm) {
    if (lp_is_identity(a)) return b;
    if (lp_is_identity(b)) return a;
    uint32_t r[4];
    lp_all_rows(lp_mul(lp_pick(k.row, a.x, b.x, a.y, b.y), lp_pick(k.row, b.zz, a.zz, b.zzz, a.zzz), k), r);   // 9 * 2 -> [2]
    const uint32_t u1 = r[0], u2 = r[1], s1 = r[2], s2 = r[3];
    const uint32_t pp_ = lp_sub<3>(u2, u1, k), rr_ = lp_sub<3>(s2, s1, k);                                   // [5], [5]
    {   // could P be a multiple of p?  P < 5 p: its low limb would be one of 0, p_0, .., 4 p_0 mod 2^29
        const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pp_) & M29;
        const uint32_t kk = (l0 * ((1u << 29) - FqParams::NINV)) & M29;
        if (kk < 5u) {
            const G1XYZZ ga = lp_to_single(a, sm), gb = lp_to_single(b, sm);
            return lp_from_single(xyzz_add(ga, gb), sm, k);
        }
    }
    lp_all_rows(lp_mul(lp_pick(k.row, pp_, rr_, a.zz, a.zzz), lp_pick(k.row, pp_, rr_, b.zz, b.zzz), k), r);   // 25 / 169 + 1 -> [2]
    const uint32_t pp = r[0], rr = r[1], zz12 = r[2], zzz12 = r[3];
    lp_all_rows(lp_mul(lp_pick(k.row, pp_, u1, zz12, zz12), pp, k), r);                                      // [2]
    const uint32_t ppp = r[0], q = r[1];
    LpPoint o;
    o.zz = r[2];
    o.x = lp_sub<7>(rr, lp_add(ppp, lp_add(q, q, k), k), k);                                                 // RR - (PPP + 2 Q [6]) + 7 p -> [9]
    const uint32_t d = lp_sub<11>(q, o.x, k), ns1 = lp_neg<3>(s1, k);                                        // [13], [3]
    lp_all_rows(lp_mul(lp_pick(k.row, rr_, ns1, zzz12, zzz12), lp_pick(k.row, d, ppp, ppp, ppp), k), r);     // 5 * 13, 3 * 2 -> [2]
    o.y = lp_add(r[0], r[1], k);                                                                            // [4]
    o.zzz = r[2];
    return o;
}

// result = sum_w 2^(c w) wsum[w]  (Horner, top window first): k_msm_final's job (msm_kernels.hpp) in the limb-parallel form,
// one wave per MSM of a batch.  Writes the XYZZ value (coordinates brought under 2 p: on-device consumers assume g1.hpp's bounds)
// and the canonical Jacobian encoding of the C ABI.
__global__ void __launch_bounds__(64) k_msm_final_lp(const uint8_t* __restrict__ wsum, int c, int W, uint8_t* __restrict__ out_xyzz,
                                                     uint8_t* __restrict__ out_jac) {
    __shared__ uint32_t sm[4 * 64];
    wsum += XYZZ_BYTES * (size_t)blockIdx.x * W;
    if (out_xyzz) out_xyzz += XYZZ_BYTES * (size_t)blockIdx.x;
    if (out_jac) out_jac += 96 * (size_t)blockIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const LpConst k = lp_const();
    LpPoint acc = lp_load(wsum + XYZZ_BYTES * (size_t)(W - 1), k);
#pragma unroll 1
    for (int w = W - 2; w >= 0; --w) {
#pragma unroll 1
        for (int i = 0; i < c; ++i) acc = lp_double(acc, k);
        acc = lp_add_points(acc, lp_load(wsum + XYZZ_BYTES * (size_t)w, k), k, sm);
    }
    G1XYZZ g = lp_to_single(acc, sm);
    if (threadIdx.x == 0) {
        if (!g.is_identity()) {   // X < 9 p, Y < 4 p here: multiply by Montgomery's 1 -> under 2 p, the value unchanged
            g.x = FQ_MUL(g.x, Fq::one());
            g.y = FQ_MUL(g.y, Fq::one());
        }
        if (out_xyzz) xyzz_store(out_xyzz, g);
        if (out_jac) jac_store_canonical(out_jac, jac_from_xyzz(g));
    }
}

}  // namespace h2agg
