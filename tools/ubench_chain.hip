// What does the one-wait-state hazard between dependent v_mad_u64_u32 (addend = result of the last-but-one instruction)
// cost?  Sequences of multiply-adds in 2 / 3 chains, with and without the s_nop the compiler inserts, at 1 and 3 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
#define M(c, a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#define NOP asm volatile("s_nop 0");
#define KERNEL(NAME, BODY, NM)                                                                        \
    __global__ void __launch_bounds__(64) NAME(uint32_t* out, uint32_t seed) {                        \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, b = seed | 1;                     \
        uint64_t c0 = a0, c1 = a1, c2 = a2;                                                           \
        for (int it = 0; it < ITERS; ++it) { BODY BODY BODY BODY }                                    \
        out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2) ^ (uint32_t)((c0 ^ c1 ^ c2) >> 32); \
    }                                                                                                 \
    static const int NAME##_mads = 4 * (NM);
KERNEL(k_2chain_nop, M(c0, a0) M(c1, a1) NOP M(c0, a1) M(c1, a2) NOP M(c0, a2) M(c1, a0) NOP, 6)
KERNEL(k_2chain_raw, M(c0, a0) M(c1, a1) M(c0, a1) M(c1, a2) M(c0, a2) M(c1, a0), 6)
KERNEL(k_3chain, M(c0, a0) M(c1, a1) M(c2, a2) M(c0, a1) M(c1, a2) M(c2, a0), 6)
KERNEL(k_1chain_nop2, M(c0, a0) asm volatile("s_nop 1"); M(c0, a1) asm volatile("s_nop 1"); M(c0, a2) asm volatile("s_nop 1");, 3)
KERNEL(k_1chain_raw, M(c0, a0) M(c0, a1) M(c0, a2), 3)
#define MS(c, a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "s"(sb) : "vcc");
#define KERNEL_S(NAME, BODY, NM)                                                                      \
    __global__ void __launch_bounds__(64) NAME(uint32_t* out, uint32_t seed) {                        \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, b = seed | 1;                     \
        const uint32_t sb = __builtin_amdgcn_readfirstlane(seed * 2654435761u | 1u);                  \
        uint64_t c0 = a0, c1 = a1, c2 = a2;                                                           \
        for (int it = 0; it < ITERS; ++it) { BODY BODY BODY BODY }                                    \
        out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2) ^ (uint32_t)((c0 ^ c1 ^ c2) >> 32) ^ b; \
    }                                                                                                 \
    static const int NAME##_mads = 4 * (NM);
KERNEL_S(k_3chain_sgpr, MS(c0, a0) MS(c1, a1) MS(c2, a2) MS(c0, a1) MS(c1, a2) MS(c2, a0), 6)
KERNEL_S(k_3chain_mixed, MS(c0, a0) M(c1, a1) MS(c2, a2) M(c0, a1) MS(c1, a2) M(c2, a0), 6)
template <class K> void run(const char* name, K k, int mads, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 1024 * wps;   // 256 CUs x 4 SIMDs x wps one-wave workgroups
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 1u); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 2u + r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        const double per_wave_ns = ms * 1e6 / ((double)ITERS * mads);   // time per multiply-add as one wave sees it
        printf("%-16s waves/SIMD %d  %.3f ms  %.2f ns per mad per wave  = %.2f ns per mad per SIMD\n", name, wps, ms, per_wave_ns, per_wave_ns / wps);
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 64 * 4096);
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k_3chain, dim3(3072), dim3(64), 0, 0, d, 1u);   // clock spin-up (~150 ms)
    hipDeviceSynchronize();
    run("2 chains + nop", k_2chain_nop, k_2chain_nop_mads, d);
    run("2 chains raw", k_2chain_raw, k_2chain_raw_mads, d);
    run("3 chains", k_3chain, k_3chain_mads, d);
    run("1 chain + nop 1", k_1chain_nop2, k_1chain_nop2_mads, d);
    run("1 chain raw", k_1chain_raw, k_1chain_raw_mads, d);
    run("3 chains, SGPR op", k_3chain_sgpr, k_3chain_sgpr_mads, d);
    run("3 chains, mixed", k_3chain_mixed, k_3chain_mixed_mads, d);
    return 0;
}
