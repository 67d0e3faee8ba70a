// Batched-affine bucket insertion, RUN ("run — not price — one batched-affine prototype").
//
// Per lane, K independent affine additions P1_k + P2_k share ONE inversion through Montgomery's trick:
//   forward   d_k = x2_k - x1_k,  pre_k = d_0 ... d_(k-1)  (kept per addition),  acc = d_0 ... d_(K-1)
//   inverse   inv = 1 / acc           (safegcd, csrc/fp.hpp fp_inv_int: the inversion every to_affine uses)
//   backward  1/d_k = inv * pre_k,  inv *= d_k,  lambda = (y2 - y1) / d_k,  x3 = lambda^2 - x1 - x2,  y3 = lambda (x1 - x3) - y1
// = 5M + 1S per addition + the inversion's share, against 8M + 2S for the XYZZ mixed addition the accumulation kernel uses.
// What decides it on this part is where pre_k lives: K x 36 B per LANE.  Two variants:
//   REG   pre_k in registers (K <= 8 fits beside the arithmetic)            -> the inversion is shared by few additions
//   MEM   pre_k in a global scratch [k][lane] (coalesced 36-B records)      -> large K, 72 B of extra traffic per addition
// Operands are read coalesced from [k][lane] arrays (a best case: the real kernel gathers 64-B records by index); the
// "points" are arbitrary field elements (the formulas do not care) and exceptional cases are not handled — this measures
// the arithmetic and the scratch traffic only, against the same harness running xyzz_add_affine.
//   hipcc --offload-arch=gfx950 -O3 -I halo2-snark-aggregator_amd/csrc tools/ubench_batched_affine.hip -o tools/ubench_batched_affine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "g1.hpp"
using namespace h2agg;

// field elements as 9 raw limbs, record i of lane l at base[(i * lanes + l) * 9 ...]  (36-byte records, lane-contiguous)
FP_INLINE Fq ld(const uint32_t* base, size_t rec) {
    Fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = base[rec * NL + i];
    return r;
}
FP_INLINE void st(uint32_t* base, size_t rec, const Fq& v) {
#pragma unroll
    for (int i = 0; i < NL; ++i) base[rec * NL + i] = v.l[i];
}

template <int K>
__global__ void __launch_bounds__(64) k_batched_affine_mem(const uint32_t* __restrict__ x1, const uint32_t* __restrict__ y1,
                                                          const uint32_t* __restrict__ x2, const uint32_t* __restrict__ y2,
                                                          uint32_t* __restrict__ scratch, uint32_t* __restrict__ out, size_t lanes) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    Fq acc = Fq::one();
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const size_t rec = (size_t)k * lanes + lane;
        const Fq d = FQ_SUB(2, ld(x2, rec), ld(x1, rec));
        st(scratch, rec, acc);                                 // pre_k
        acc = FQ_MUL(acc, d);
    }
    Fq inv = fp_inv<FqParams>(acc);
    Fq sx = Fq::zero(), sy = Fq::zero();
#pragma unroll 1
    for (int k = K - 1; k >= 0; --k) {
        const size_t rec = (size_t)k * lanes + lane;
        const Fq a1 = ld(x1, rec), b1 = ld(y1, rec), a2 = ld(x2, rec), b2 = ld(y2, rec);
        const Fq d = FQ_SUB(2, a2, a1);
        const Fq invd = FQ_MUL(inv, ld(scratch, rec));
        inv = FQ_MUL(inv, d);
        const Fq lam = FQ_MUL(FQ_SUB(2, b2, b1), invd);
        const Fq x3 = fp_sub_sub2<6, FqParams>(FQ_SQR(lam), FQ_ADD(a1, a2), Fq::zero());
        const Fq y3 = FQ_SUB(2, FQ_MUL(lam, FQ_SUB(8, a1, x3)), b1);
        sx = FQ_ADD(fp_cond_sub<FqParams>(fp_mul<FqParams>(sx, Fq::one())), x3);   // (keeps the results live; folded so bounds hold)
        sy = FQ_ADD(fp_cond_sub<FqParams>(fp_mul<FqParams>(sy, Fq::one())), y3);
    }
    st(out, lane, sx);
    st(out, lanes + lane, sy);
}

template <int K>
__global__ void __launch_bounds__(64) k_batched_affine_reg(const uint32_t* __restrict__ x1, const uint32_t* __restrict__ y1,
                                                          const uint32_t* __restrict__ x2, const uint32_t* __restrict__ y2,
                                                          uint32_t* __restrict__ out, size_t lanes, int rounds) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    Fq sx = Fq::zero(), sy = Fq::zero();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        Fq pre[K];
        Fq acc = Fq::one();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t rec = (size_t)(r * K + k) * lanes + lane;
            const Fq d = FQ_SUB(2, ld(x2, rec), ld(x1, rec));
            pre[k] = acc;
            acc = FQ_MUL(acc, d);
        }
        Fq inv = fp_inv<FqParams>(acc);
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            const size_t rec = (size_t)(r * K + k) * lanes + lane;
            const Fq a1 = ld(x1, rec), b1 = ld(y1, rec), a2 = ld(x2, rec), b2 = ld(y2, rec);
            const Fq d = FQ_SUB(2, a2, a1);
            const Fq invd = FQ_MUL(inv, pre[k]);
            inv = FQ_MUL(inv, d);
            const Fq lam = FQ_MUL(FQ_SUB(2, b2, b1), invd);
            const Fq x3 = fp_sub_sub2<6, FqParams>(FQ_SQR(lam), FQ_ADD(a1, a2), Fq::zero());
            const Fq y3 = FQ_SUB(2, FQ_MUL(lam, FQ_SUB(8, a1, x3)), b1);
            sx = FQ_ADD(fp_cond_sub<FqParams>(fp_mul<FqParams>(sx, Fq::one())), x3);
            sy = FQ_ADD(fp_cond_sub<FqParams>(fp_mul<FqParams>(sy, Fq::one())), y3);
        }
    }
    st(out, lane, sx);
    st(out, lanes + lane, sy);
}

// the same harness with the production formula: `total` mixed additions per lane into one XYZZ accumulator
__global__ void __launch_bounds__(64) k_xyzz_mixed(const uint32_t* __restrict__ x2, const uint32_t* __restrict__ y2,
                                                  uint32_t* __restrict__ out, size_t lanes, int total) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    G1XYZZ acc = G1XYZZ::identity();
#pragma unroll 1
    for (int k = 0; k < total; ++k) {
        const size_t rec = (size_t)k * lanes + lane;
        G1Affine q;
        q.x = ld(x2, rec);
        q.y = ld(y2, rec);
        xyzz_add_affine(acc, q);
    }
    st(out, lane, acc.x);
    st(out, lanes + lane, acc.zz);
}

int main() {
    const size_t lanes = 3072 * 64;      // 3 one-wave workgroups per SIMD
    const int TOTAL = 64;                // additions per lane in every variant
    const size_t recs = (size_t)TOTAL * lanes;
    std::vector<uint32_t> h(recs * NL);
    uint32_t* d[4];
    uint64_t s = 88172645463325252ull;
    for (int a = 0; a < 4; ++a) {
        for (size_t i = 0; i < h.size(); ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[i] = (uint32_t)s & ((i % NL == NL - 1) ? 0xfffffu : M29);   // tight limbs, value < p
        }
        hipMalloc(&d[a], h.size() * 4);
        hipMemcpy(d[a], h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    uint32_t *scratch, *out;
    hipMalloc(&scratch, recs * NL * 4);
    hipMalloc(&out, 2 * lanes * 16 * NL * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time_it = [&](const char* name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        // 3 waves per SIMD worth of lanes: additions per SIMD = 3 * 64 lanes... report time per addition per LANE-slot and
        // the whole-chip rate
        printf("%-34s %8.3f ms   %7.1f ns per lock-step addition of all lanes   %6.2f G additions/s\n", name, ms, ms * 1e6 / TOTAL,
               (double)recs / ms / 1e6);
    };
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(k_xyzz_mixed, dim3(lanes / 64), dim3(64), 0, 0, d[2], d[3], out, lanes, TOTAL);   // spin-up
    time_it("xyzz mixed addition (production)", [&] { hipLaunchKernelGGL(k_xyzz_mixed, dim3(lanes / 64), dim3(64), 0, 0, d[2], d[3], out, lanes, TOTAL); });
    time_it("batched affine, REG, K = 4", [&] { hipLaunchKernelGGL(k_batched_affine_reg<4>, dim3(lanes / 64), dim3(64), 0, 0, d[0], d[1], d[2], d[3], out, lanes, TOTAL / 4); });
    time_it("batched affine, REG, K = 8", [&] { hipLaunchKernelGGL(k_batched_affine_reg<8>, dim3(lanes / 64), dim3(64), 0, 0, d[0], d[1], d[2], d[3], out, lanes, TOTAL / 8); });
    // (fewer additions per lane = more lanes for the same total: L = lanes * TOTAL / K)
    time_it("batched affine, MEM, K = 16", [&] { hipLaunchKernelGGL(k_batched_affine_mem<16>, dim3(lanes * 4 / 64), dim3(64), 0, 0, d[0], d[1], d[2], d[3], scratch, out, lanes * 4); });
    time_it("batched affine, MEM, K = 32", [&] { hipLaunchKernelGGL(k_batched_affine_mem<32>, dim3(lanes * 2 / 64), dim3(64), 0, 0, d[0], d[1], d[2], d[3], scratch, out, lanes * 2); });
    time_it("batched affine, MEM, K = 64", [&] { hipLaunchKernelGGL(k_batched_affine_mem<64>, dim3(lanes / 64), dim3(64), 0, 0, d[0], d[1], d[2], d[3], scratch, out, lanes); });
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k_batched_affine_mem<64>);
    printf("k_batched_affine_mem<64>: %d VGPRs; ", fa.numRegs);
    hipFuncGetAttributes(&fa, (const void*)k_batched_affine_reg<8>);
    printf("k_batched_affine_reg<8>: %d VGPRs, %zu B scratch; ", fa.numRegs, (size_t)fa.localSizeBytes);
    hipFuncGetAttributes(&fa, (const void*)k_xyzz_mixed);
    printf("k_xyzz_mixed: %d VGPRs\n", fa.numRegs);
    return 0;
}
