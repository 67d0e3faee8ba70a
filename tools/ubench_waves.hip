// Does the field arithmetic of k_msm_accumulate issue faster at FOUR waves per SIMD than at three?
// ("an odd wave count loses ~16 %" — that figure came from bare v_mad_u64_u32 chains, tools/ubench_chain.hip.)
// This runs the real instruction mix — csrc/fp.hpp's two-chain Montgomery products with their carry sweeps, subtractions and
// quotient digits, no memory traffic — as one-wave workgroups at 1..6 waves per SIMD (occupancy capped with unused LDS, the way
// H2AGG_ACC_LDS does it for the kernel itself) and prints the time per field multiplication per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -I halo2-snark-aggregator_amd/csrc tools/ubench_waves.hip -o tools/ubench_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "fp.hpp"
using namespace h2agg;
typedef Fp<FqParams> Fq_;
#define ITERS 600
// per iteration: 2 dual products + 2 dual squarings (8 multiplications' worth, the mix of a mixed addition) + 2 subtractions
__global__ void __launch_bounds__(64) k_mix(uint32_t* out, uint32_t seed) {
    extern __shared__ uint32_t lds_pad[];
    Fq_ a, b, c, d;
    for (int i = 0; i < NL; ++i) {
        a.l[i] = (seed * 2654435761u + threadIdx.x * 40503u + i) & M29;
        b.l[i] = (seed * 40503u + threadIdx.x * 2654435761u + 7 * i) & M29;
        c.l[i] = (a.l[i] * 3 + 1) & M29;
        d.l[i] = (b.l[i] * 5 + 2) & M29;
    }
    a.l[8] &= 0xfffff; b.l[8] &= 0xfffff; c.l[8] &= 0xfffff; d.l[8] &= 0xfffff;
    for (int it = 0; it < ITERS; ++it) {
        Fq_ u, v, w, x;
        fp_mul_dual<FqParams>(a, b, c, d, u, v);
        Fq_ p = fp_sub<4, FqParams>(u, c);
        Fq_ r = fp_sub<4, FqParams>(v, a);
        fp_sqr_dual<FqParams>(p, r, w, x);
        fp_mul_dual<FqParams>(w, p, x, r, a, c);
        fp_sqr_dual<FqParams>(u, v, b, d);
    }
    uint32_t o = 0;
    for (int i = 0; i < NL; ++i) o ^= a.l[i] ^ b.l[i] ^ c.l[i] ^ d.l[i];
    if (seed == 0xdeadbeef) lds_pad[threadIdx.x] = o;
    out[blockIdx.x * 64 + threadIdx.x] = o;
}
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 64 * 16384);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k_mix);
    printf("k_mix: %d VGPRs, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
    hipFuncSetAttribute((const void*)k_mix, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(k_mix, dim3(3072), dim3(64), 0, 0, d, 1u);   // clock spin-up
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)   // 0: occupancy capped with unused LDS, two rounds; 1: no cap, exactly 1024 x wps one-wave workgroups (one round)
    for (int wps : {1, 2, 3, 4, 5, 6}) {
        // 160 KiB of LDS per CU: a one-wave workgroup holding 160 KiB / (4 wps) leaves room for exactly 4 wps of them
        const size_t lds = (mode == 1 || wps >= 6) ? 0 : (160 * 1024 / (4 * wps)) - 512;
        const int rounds = mode == 0 ? 2 : 1;
        const int blocks = 1024 * wps * rounds;
        hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(64), lds, 0, d, 1u); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(64), lds, 0, d, 2u + r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        const double mults = 8.0 * ITERS * rounds;   // per wave slot
        printf("%s waves/SIMD %d  lds %6zu  %.3f ms  %.1f ns per multiplication per SIMD (wave64)\n", mode ? "free  " : "capped", wps, lds, ms, ms * 1e6 / mults / wps);
    }
    return 0;
}
