// Sustained (power-capped) rates of the two candidate multiplier instructions on the whole chip:
//   v_mad_u64_u32 (32 x 32 + 64 -> 64: what fp.hpp's 9 x 29-bit Montgomery product is made of, 162 per product) and
//   v_fma_f64     (53-bit mantissa: the double-precision route to a 5 x 52-bit product, ~100 per product)
// Every SIMD holds four waves of eight independent chains; each kernel runs ~0.5 s so that the package sits at its power cap
// (rocm-smi: 1 390 W of 1 400 during the MSM loop).  Prints giga-instructions per second (wave64 instructions x 64 lanes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_power.hip -o tools/ubench_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define MAD(c, a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
__global__ void __launch_bounds__(256) k_mad(uint64_t* out, uint32_t seed, int iters) {
    uint32_t a0 = seed * 2654435761u + threadIdx.x * 40503u, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3, a4 = a0 * 9 + 11, a5 = a0 * 11 + 5,
             a6 = a0 * 13 + 9, a7 = a0 * 17 + 1, b = (a0 >> 3) | 0x10000001u;
    uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { MAD(c0, a0) MAD(c1, a1) MAD(c2, a2) MAD(c3, a3) MAD(c4, a4) MAD(c5, a5) MAD(c6, a6) MAD(c7, a7) }
        a0 ^= (uint32_t)(c0 >> 17); a1 ^= (uint32_t)(c1 >> 19);   // (keep the operands changing: toggling data is what draws power)
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
#define FMA(c, a) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(y));
__global__ void __launch_bounds__(256) k_fma(uint64_t* out, uint32_t seed, int iters) {
    // operands with full 52-bit mantissas near 1; accumulators bounded by multiplying with values < 1 in magnitude alternately
    double x0 = 1.0 + 1e-7 * (seed % 97 + threadIdx.x) + 3.141592653589793e-11, x1 = -x0 * 0.999999, x2 = x0 * 0.87654321, x3 = -x0 * 0.7654321,
           x4 = x0 * 0.654321, x5 = -x0 * 0.54321, x6 = x0 * 0.4321, x7 = -x0 * 0.321, y = 0.99999999 - 1e-9 * threadIdx.x;
    double c0 = 1.1, c1 = 2.2, c2 = 3.3, c3 = 4.4, c4 = 5.5, c5 = 6.6, c6 = 7.7, c7 = 8.8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { FMA(c0, x0) FMA(c1, x1) FMA(c2, x2) FMA(c3, x3) FMA(c4, x4) FMA(c5, x5) FMA(c6, x6) FMA(c7, x7) }
        c0 *= 0.5; c1 *= 0.5;   // (bounded)
    }
    out[blockIdx.x * 256 + threadIdx.x] = __double_as_longlong(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
}
// the integer matrix cores for comparison (no field arithmetic is built on them: see DESIGN.md section 9): 32 x 32 x 16 int8 MACs
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_mfma(uint64_t* out, uint32_t seed, int iters) {
    long a = 0x0102030405060708L * (long)(seed | 1) + threadIdx.x * 0x0101010101010101L, b = 0x1112131415161718L ^ (long)threadIdx.x * 0x0303030303030303L;
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(b, b, c3, 0, 0, 0);
        a ^= (long)c0[0];   // (operands keep changing)
        b += 0x0101010101010101L;
    }
    uint64_t acc = 0;
    for (int i = 0; i < 16; ++i) acc += (uint32_t)(c0[i] ^ c1[i] ^ c2[i] ^ c3[i]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 4;   // four 256-thread workgroups per CU: four waves per SIMD
    uint64_t* out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 1500000;
    for (int round = 0; round < 3; ++round) {
        for (int which = 0; which < 2; ++which) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, out, 7u + round, iters);
            else hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 7u + round, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)blocks * 4 * 32.0 * iters;   // wave64 instructions
            if (which == 0 && round == 0) {   // once: the matrix cores, 4 MFMAs per iteration
                hipEventRecord(e0);
                const int mi = iters / 4;
                hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, 3u, mi);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float mms = 0;
                hipEventElapsedTime(&mms, e0, e1);
                const double mf = (double)blocks * 4 * 4.0 * mi;
                printf("v_mfma_i32_32x32x16_i8  %.0f ms   %.2f G wave-instructions/s = %.0f T int8 MACs/s = %.2e 1-bit products/s  (v_mad_u64_u32 at 29 x 29 bits: see below x 64 x 841)\n",
                       mms, mf / (mms * 1e-3) / 1e9, mf * 16384.0 / (mms * 1e-3) / 1e12, mf * 16384.0 * 64.0 / (mms * 1e-3));
            }
            printf("%s  %.0f ms   %.1f G wave-instructions/s   %.2f cycles per instruction per SIMD at 2.4 GHz\n", which ? "v_fma_f64    " : "v_mad_u64_u32",
                   ms, insts / (ms * 1e-3) / 1e9, 2.4e9 * (ms * 1e-3) / (insts / (prop.multiProcessorCount * 4)));
        }
    }
    return 0;
}
