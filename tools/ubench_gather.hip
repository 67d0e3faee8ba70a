// Calibration of rocprofv3's FETCH_SIZE for THIS path's read pattern (MI355X_MICROARCH.md: the counter reports half the bytes
// of a wide coalesced stream; other patterns are to be calibrated on a known byte count): every lane gathers whole 64-byte
// records (4 x global_load_dwordx4) at pseudo-random indices, as k_msm_accumulate does with its bases.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o tools/ubench_gather
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench_gather      (bytes gathered are printed for comparison)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k_gather64(const uint4* __restrict__ table, uint32_t mask, uint32_t per_thread, uint4* out) {
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint4* r = table + 4 * (size_t)((x >> 4) & mask);
        const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        acc.y += a.y + b.z + c.w + d.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_stream16(const uint4* __restrict__ table, size_t n, uint4* out) {   // the guide's reference pattern
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 a = table[i];
        acc.x ^= a.x; acc.y += a.y; acc.z ^= a.z; acc.w += a.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const size_t records_big = (size_t)1 << 24, records_small = (size_t)1 << 20;   // 1 GiB (beyond the 256-MiB Infinity Cache), 64 MiB (the MSM's base table)
    uint4 *table, *out;
    hipMalloc(&table, records_big * 64);
    hipMalloc(&out, 4096 * 256 * 16);
    hipMemset(table, 1, records_big * 64);
    const uint32_t blocks = 4096, per_thread = 16;   // 2^24 gathers of 64 B = 1 GiB per launch
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(256), 0, 0, (const uint4*)table, (uint32_t)(records_big - 1), per_thread, out);
        hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(256), 0, 0, (const uint4*)table, (uint32_t)(records_small - 1), per_thread, out);
        hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(256), 0, 0, (const uint4*)table, records_big * 4, out);
    }
    hipDeviceSynchronize();
    printf("k_gather64: %zu bytes gathered per launch (first launch of each pair: 1-GiB table, second: 64-MiB table); k_stream16: %zu bytes streamed\n",
           (size_t)blocks * 256 * per_thread * 64, records_big * 64);
    return 0;
}
