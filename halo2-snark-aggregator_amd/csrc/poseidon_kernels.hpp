// Poseidon sponge + transcript on the device.
//
// Stands behind
//   PoseidonChip::{update, squeeze, permutation}   halo2-snark-aggregator-api/src/hash/poseidon.rs:167-230
//       absorb_with_pre_constants :45-86 (padding one), x_power5_with_constant :9-19, apply_mds :88-110,
//       apply_sparse_mds :112-141
//   PoseidonEncode::{encode_point, encode_scalar}  halo2-snark-aggregator-api/src/mock/transcript_encode.rs:28-63
//   PoseidonTranscriptRead::{read_point, read_scalar, common_point, common_scalar, squeeze_challenge_scalar}
//                                                  halo2-snark-aggregator-api/src/systems/halo2/transcript.rs:56-179
// with T = 9, RATE = 8, R_F = 8, R_P = 63 (halo2-snark-aggregator-circuit/src/verify_circuit.rs:127-135).
//
// A transcript is a strictly sequential chain (state_{k+1} = permutation(state_k + chunk_k)); what is parallel is the
// batch: every proof of an aggregation has its own sponge.  One group of 16 lanes per proof (lanes 0..8 hold the nine
// state words, Montgomery, < 2r), four proofs per 64-lane workgroup; all proofs of a launch share the layout, so control
// flow is uniform and the lanes of a group exchange words through LDS with workgroup barriers.  The reader's interleaved
// "read, absorb, squeeze" becomes: the host lays out the element stream once per circuit (which proof items are points,
// which scalars, after how many elements each challenge is squeezed), one kernel builds every proof's elements from the
// decompressed points and the raw scalars, and one kernel runs the sponges.
//
// Constants: csrc/poseidon_host.hpp (Grain LFSR + optimized schedule), uploaded once per context as 9-limb registers:
//   START(k, i) k < 5 | PARTIAL(k) k < 63 | END(k, i) k < 3 | MDS(i, j) | PRE(i, j) | SROW(k, j) | SCOL(k, j) j < 8 | FIN(i)
// (SROW / SCOL / FIN hold the SCALED partial rounds: D_k | R_k, A_k, beta_63 | cum — poseidon_host.hpp scale_partial_rounds)
#pragma once
#include "schema.hpp"

namespace h2agg {

constexpr int PSD_T = 9, PSD_RF = 8, PSD_RP = 63, PSD_H = PSD_RF / 2;
constexpr uint32_t PSD_START = 0, PSD_PARTIAL = PSD_START + (PSD_H + 1) * PSD_T, PSD_END = PSD_PARTIAL + PSD_RP,
                   PSD_MDS = PSD_END + (PSD_H - 1) * PSD_T, PSD_PRE = PSD_MDS + PSD_T * PSD_T,
                   PSD_SROW = PSD_PRE + PSD_T * PSD_T, PSD_SCOL = PSD_SROW + PSD_RP * PSD_T,
                   PSD_FIN = PSD_SCOL + PSD_RP * (PSD_T - 1), PSD_NCONST = PSD_FIN + PSD_T;
constexpr int PSD_GROUP = 16, PSD_BLOCK = 64, PSD_PER_BLOCK = PSD_BLOCK / PSD_GROUP;

FP_INLINE Fr fr_add2r(const Fr& a, const Fr& b) { return fr_fold_2r(fp_add<FrParams>(a, b)); }   // < 2r + < 2r -> < 2r
FP_INLINE Fr fr_pow5_plus(const Fr& x, const Fr& c) {   // x^5 + c  (x_power5_with_constant, poseidon.rs:9-19)
    const Fr x2 = fp_sqr<FrParams>(x);
    const Fr x4 = fp_sqr<FrParams>(x2);
    return fr_add2r(fp_mul<FrParams>(x4, x), c);
}

struct PsdLds {
    uint32_t w[PSD_PER_BLOCK][PSD_T + 1][NL];   // slots 0..8: one word per lane; slot 9: the broadcast s0 of a partial round
};
FP_INLINE void psd_put(PsdLds& x, int g, int i, const Fr& v) {
#pragma unroll
    for (int k = 0; k < NL; ++k) x.w[g][i][k] = v.l[k];
}
FP_INLINE Fr psd_get(const PsdLds& x, int g, int i) {
    Fr v;
#pragma unroll
    for (int k = 0; k < NL; ++k) v.l[k] = x.w[g][i][k];
    return v;
}
// lane 0 of every 16-lane group -> all lanes of the group (ds_bpermute; the groups are DPP rows)
FP_INLINE Fr psd_bcast0(const Fr& v) {
    Fr r;
    const int src = (int)(threadIdx.x & ~(PSD_GROUP - 1)) & 63;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.l[k] = (uint32_t)__shfl((int)v.l[k], src, 64);
    return r;
}
// sum over the 16 lanes of a group, result valid in lane 0: four row_shl steps (lanes past the row read zero)
template <int N>
FP_INLINE Fr psd_shl(const Fr& v) {
    Fr r;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.l[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[k], 0x100 + N, 0xF, 0xF, true);
    return r;
}
// Inputs < 2r on at most nine lanes (zero elsewhere); the sum is left unfolded, < 18r: it is only ever an operand of the next
// round's multiplications (18 * 18 / 169 + 1 < 3), and every fold on this chain is ~30 instructions of a lone wave's time.
FP_INLINE Fr psd_row_sum(Fr t) {
    t = fp_add<FrParams>(t, psd_shl<1>(t));
    t = fp_add<FrParams>(t, psd_shl<2>(t));
    t = fp_add<FrParams>(t, psd_shl<4>(t));
    return fp_add<FrParams>(t, psd_shl<8>(t));
}
// s <- M s for a dense T x T matrix at spec[base ..]: every lane publishes its word, then forms its row's dot product as
// three 3-term products with one Montgomery reduction each (fp_mul3)
FP_INLINE Fr psd_dense(const uint32_t* __restrict__ spec, uint32_t base, PsdLds& x, int g, int l, int lc, const Fr& s) {
    if (l < PSD_T) psd_put(x, g, l, s);
    __syncthreads();
    const uint32_t rb = base + lc * PSD_T;
    Fr acc = fp_mul3_ps<FrParams>(reg_load(spec, rb), psd_get(x, g, 0), reg_load(spec, rb + 1), psd_get(x, g, 1),
                                  reg_load(spec, rb + 2), psd_get(x, g, 2));
    acc = fr_add2r(acc, fp_mul3_ps<FrParams>(reg_load(spec, rb + 3), psd_get(x, g, 3), reg_load(spec, rb + 4), psd_get(x, g, 4),
                                             reg_load(spec, rb + 5), psd_get(x, g, 5)));
    acc = fr_add2r(acc, fp_mul3_ps<FrParams>(reg_load(spec, rb + 6), psd_get(x, g, 6), reg_load(spec, rb + 7), psd_get(x, g, 7),
                                             reg_load(spec, rb + 8), psd_get(x, g, 8)));
    __syncthreads();
    return acc;
}

// PoseidonChip::permutation (poseidon.rs:193-230) on the group's state; `inp` = this lane's absorbed input (Montgomery;
// zero on lanes without one), nin = inputs in this chunk (< T)
FP_INLINE Fr psd_permute(const uint32_t* __restrict__ spec, PsdLds& x, int g, int l, int lc, Fr s, const Fr& inp, uint32_t nin) {
    // absorb_with_pre_constants: s[0] += pc[0]; s[1 ..= nin] += input + pc; s[nin + 1] += pc + 1; the rest += pc
    s = fr_add2r(s, reg_load(spec, PSD_START + lc));
    s = fr_add2r(s, inp);
    if ((uint32_t)l == nin + 1) s = fr_add2r(s, Fr::one());
#pragma unroll 1
    for (int k = 1; k < PSD_H; ++k) {
        s = fr_pow5_plus(s, reg_load(spec, PSD_START + k * PSD_T + lc));
        s = psd_dense(spec, PSD_MDS, x, g, l, lc, s);
    }
    s = fr_pow5_plus(s, reg_load(spec, PSD_START + PSD_H * PSD_T + lc));
    s = psd_dense(spec, PSD_PRE, x, g, l, lc, s);
    if (l >= PSD_T) s = Fr::zero();   // idle lanes carry zeros: they take part in the row sums below
    // Partial rounds, scaled (Spec::scale_partial_rounds): lane 0 carries w (s0 = beta_k w), lanes 1..8 carry shat_i.
    // THREE multiplication issues per round for the whole group (a wave's lanes multiply in lockstep):
    //   1: lane 0: w^2            lanes >= 1: A_{k-1,i} * z_{k-1}   (the update the previous round owes them)
    //   2: lane 0: w^4            lanes >= 1: R_{k,i} * shat_i
    //   3: lane 0: z_k = w^5      lanes >= 1: —
    // then w <- z_k + D_k + sum_i R_{k,i} shat_i.  This round's constants were loaded during the previous round.
    Fr zprev = Fr::zero(), aprev = Fr::zero();
    Fr rk = reg_load(spec, PSD_SROW + lc);                       // lane 0: D_0; lanes 1..8: R_{0,i}
    Fr ak = reg_load(spec, PSD_SCOL + (lc ? lc - 1 : 0));        // A_{0,i}
#pragma unroll 1
    for (int k = 0; k < PSD_RP; ++k) {
        const int kn = k + 1 < PSD_RP ? k + 1 : k;
        const Fr rk_n = reg_load(spec, PSD_SROW + kn * PSD_T + lc);
        const Fr ak_n = reg_load(spec, PSD_SCOL + kn * (PSD_T - 1) + (lc ? lc - 1 : 0));
        Fr a1, b1;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            a1.l[i] = (l == 0) ? s.l[i] : aprev.l[i];
            b1.l[i] = (l == 0) ? s.l[i] : zprev.l[i];
        }
        const Fr p1 = fp_mul<FrParams>(a1, b1);                  // lane 0: w^2 (< 4r); others: A z (zero in round 0)
        const Fr sh = fr_fold_2r(fp_add<FrParams>(s, p1));       // lanes >= 1: shat_i + A z, < 2r
        Fr a2, b2;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            a2.l[i] = (l == 0) ? p1.l[i] : rk.l[i];
            b2.l[i] = (l == 0) ? p1.l[i] : sh.l[i];
        }
        const Fr p2 = fp_mul<FrParams>(a2, b2);                  // lane 0: w^4; others: R shat_i
        const Fr z = fp_mul<FrParams>(p2, s);                    // lane 0: w^5 (others: unused)
        Fr t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.l[i] = (l >= 1 && l < PSD_T) ? p2.l[i] : 0u;
        const Fr sum = psd_row_sum(t);                           // lane 0: sum_i R shat_i, < 16r
        const Fr wn = fp_add<FrParams>(fp_add<FrParams>(z, rk), sum);   // lane 0: z + D_k + sum, < 20r: only ever squared / multiplied
        zprev = psd_bcast0(z);
#pragma unroll
        for (int i = 0; i < NL; ++i) s.l[i] = (l == 0) ? wn.l[i] : (l < PSD_T ? sh.l[i] : 0u);
        aprev = ak;
        rk = rk_n;
        ak = ak_n;
    }
    {   // the last update owed to lanes 1..8, the constants they were carried without, and s0 = beta_63 w
        const Fr fin = reg_load(spec, PSD_FIN + lc);
        Fr a1, b1;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            a1.l[i] = (l == 0) ? fin.l[i] : aprev.l[i];
            b1.l[i] = (l == 0) ? s.l[i] : zprev.l[i];
        }
        const Fr p1 = fp_mul<FrParams>(a1, b1);                  // lane 0: beta_63 w (< 2r); others: A z
        const Fr up = fr_fold_2r(fp_add<FrParams>(fr_fold_2r(fp_add<FrParams>(s, p1)), fin));
#pragma unroll
        for (int i = 0; i < NL; ++i) s.l[i] = (l == 0) ? p1.l[i] : (l < PSD_T ? up.l[i] : 0u);
    }
#pragma unroll 1
    for (int k = 0; k < PSD_H - 1; ++k) {
        s = fr_pow5_plus(s, reg_load(spec, PSD_END + k * PSD_T + lc));
        s = psd_dense(spec, PSD_MDS, x, g, l, lc, s);
    }
    s = fr_pow5_plus(s, Fr::zero());
    return psd_dense(spec, PSD_MDS, x, g, l, lc, s);
}

// One sponge per proof.  elems: canonical Fr (32 B) at elems[p * elem_stride + 32 * i]; upto[q] = number of elements
// absorbed before squeeze q (non-decreasing; equal consecutive values = squeeze again without absorbing); out[p][q] =
// challenge q, canonical.  PoseidonChip::squeeze (poseidon.rs:171-191): pending elements in chunks of RATE, one more
// permutation of the empty chunk when the last chunk was full or there was nothing to absorb.
__global__ void __launch_bounds__(PSD_BLOCK) k_poseidon_transcript(const uint32_t* __restrict__ spec,
                                                                   const uint8_t* __restrict__ elems, size_t elem_stride,
                                                                   const uint32_t* __restrict__ upto, uint32_t nsq,
                                                                   uint32_t nproofs, uint8_t* __restrict__ out,
                                                                   uint32_t* flags) {
    __shared__ PsdLds x;
    const int g = threadIdx.x / PSD_GROUP, l = threadIdx.x % PSD_GROUP, lc = l < PSD_T ? l : PSD_T - 1;
    const uint32_t p = blockIdx.x * PSD_PER_BLOCK + g;
    const bool live = p < nproofs;
    const uint8_t* my = elems + (size_t)(live ? p : nproofs - 1) * elem_stride;
    Fr s = Fr::zero();
    if (l == 0) {   // poseidon::State::default(): (2^64, 0, ..., 0)
        Fr v = Fr::zero();
        v.l[2] = 1u << 6;   // 2^64 = 2^(2*29 + 6)
        s = fp_to_mont<FrParams>(v);
    }
    uint32_t pos = 0, bad = 0;
#pragma unroll 1
    for (uint32_t q = 0; q < nsq; ++q) {
        const uint32_t end = upto[q];
        uint32_t padding_offset = 0;
        bool any = false;
#pragma unroll 1
        while (pos < end) {
            const uint32_t nin = (end - pos < (uint32_t)(PSD_T - 1)) ? end - pos : (uint32_t)(PSD_T - 1);
            Fr inp = Fr::zero();
            if (l >= 1 && (uint32_t)l <= nin) {
                const Fr c = fp_load<FrParams>(my + 32 * (size_t)(pos + l - 1));
                bad |= !fp_is_canonical<FrParams>(c);
                inp = fp_to_mont<FrParams>(c);
            }
            s = psd_permute(spec, x, g, l, lc, s, inp, nin);
            padding_offset = (uint32_t)(PSD_T - 1) - nin;
            pos += nin;
            any = true;
        }
        if (!any || padding_offset == 0) s = psd_permute(spec, x, g, l, lc, s, Fr::zero(), 0);
        if (live && l == 1) fp_store<FrParams>(out + 32 * ((size_t)p * nsq + q), fp_from_mont<FrParams>(s));
    }
    if (bad && live) atomicOr(flags, FLAG_NONCANONICAL);
}

// ---- the element stream of a proof's transcript -----------------------------------------------------------------------
// item kinds of a layout entry
enum : uint32_t { TR_CONST = 0, TR_POINT_EXT = 1, TR_POINT = 2, TR_SCALAR = 3 };
struct TrItem {
    uint32_t kind;   // TR_*
    uint32_t src;    // CONST: index into consts; POINT_EXT: index into the proof's external points (instance commitments);
                     // POINT / SCALAR: byte offset of the 32-byte item inside the proof
    uint32_t dst;    // first element index written (points write two)
    uint32_t pidx;   // POINT: index of the decompressed point among the proof's points
};
FP_INLINE Fr fq_int_mod_r(const Fq& a) {   // canonical integer < p  ->  canonical integer mod r  (p < 2r)
    int32_t d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (int32_t)a.l[i] - (int32_t)FrParams::MOD[i];
    Fr t = fp_normalize<FrParams>(d);
    const bool neg = (int32_t)t.l[8] < 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = neg ? a.l[i] : t.l[i];
    return r;
}
// decompress the proofs' points (C::from_bytes, transcript.rs:63-70) and lay out every proof's element stream:
// PoseidonEncode: a point -> (x mod r, y mod r), identity -> (0, 0); a scalar -> itself (Fr::from_repr rejects >= r:
// "invalid field element encoding in proof" -> FLAG_NONCANONICAL).  One lane per (proof, item).
__global__ void __launch_bounds__(BLOCK) k_transcript_elements(const uint8_t* __restrict__ proofs, size_t proof_stride,
                                                               const uint8_t* __restrict__ ext_points, uint32_t n_ext,
                                                               const uint8_t* __restrict__ consts,
                                                               const TrItem* __restrict__ items, uint32_t nitems,
                                                               uint32_t nproofs, uint32_t npoints,
                                                               uint8_t* __restrict__ points_out /* [proof][npoints][64] */,
                                                               uint8_t* __restrict__ elems, size_t elem_stride,
                                                               uint32_t* flags, int which = 0) {
    // which: 0 = every item; 1 = everything but the external points (what depends on the proof bytes only: runs beside the
    // instance-column MSMs that produce the external points); 2 = the external points only
    const size_t total = (size_t)nproofs * nitems;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (size_t)gridDim.x * BLOCK) {
        const uint32_t p = (uint32_t)(t / nitems);
        const TrItem it = items[t % nitems];
        if ((which == 1 && it.kind == TR_POINT_EXT) || (which == 2 && it.kind != TR_POINT_EXT)) continue;
        uint8_t* e = elems + (size_t)p * elem_stride + 32 * (size_t)it.dst;
        if (it.kind == TR_CONST) {
            const Fr c = fp_load<FrParams>(consts + 32 * (size_t)it.src);
            if (!fp_is_canonical<FrParams>(c)) atomicOr(flags, FLAG_NONCANONICAL);
            fp_store<FrParams>(e, c);
        } else if (it.kind == TR_SCALAR) {
            const Fr c = fp_load<FrParams>(proofs + (size_t)p * proof_stride + it.src);
            if (!fp_is_canonical<FrParams>(c)) atomicOr(flags, FLAG_NONCANONICAL);
            fp_store<FrParams>(e, c);
        } else {
            Fq ox, oy;
            if (it.kind == TR_POINT_EXT) {
                const uint8_t* src = ext_points + 64 * ((size_t)p * n_ext + it.src);
                ox = fp_load<FqParams>(src);
                oy = fp_load<FqParams>(src + 32);
                if (!fp_is_canonical<FqParams>(ox) | !fp_is_canonical<FqParams>(oy)) atomicOr(flags, FLAG_NONCANONICAL);
            } else {
                const bool good = g1_decompress_one(u256_load(proofs + (size_t)p * proof_stride + it.src), ox, oy);
                if (!good) atomicOr(flags, FLAG_BAD_POINT);
                uint8_t* po = points_out + 64 * ((size_t)p * npoints + it.pidx);
                fp_store<FqParams>(po, ox);
                fp_store<FqParams>(po + 32, oy);
            }
            fp_store<FrParams>(e, fq_int_mod_r(ox));
            fp_store<FrParams>(e + 32, fq_int_mod_r(oy));
        }
    }
}

}  // namespace h2agg
