// Lone-wave latency of the serial-tail building blocks (k_msm_final is one wave): dependent chains of fq_mul,
// xyzz_double (one lane), xyzz_double_par4 (4 lanes cooperating) and xyzz_add, timed with HIP events, plus the
// shader clock the wave actually saw (clock64 ticks / wall_clock64 time).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_latency.hip -o tools/ubench_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../halo2-snark-aggregator_amd/csrc/msm_kernels.hpp"
using namespace h2agg;

__global__ void __launch_bounds__(64) k_chain(int mode, int iters, const uint8_t* in, uint8_t* out, uint64_t* clk) {
    G1XYZZ p = xyzz_load(in);
    G1XYZZ q = xyzz_load(in + XYZZ_BYTES);
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    if (mode == 0) {
        Fq a = p.x;
#pragma unroll 1
        for (int i = 0; i < iters; ++i) a = FQ_MUL(a, p.y);
        p.x = a;
    } else if (mode == 1) {
#pragma unroll 1
        for (int i = 0; i < iters; ++i) p = xyzz_double(p);
    } else if (mode == 2) {
#pragma unroll 1
        for (int i = 0; i < iters; ++i) p = xyzz_double_par4(p);
    } else if (mode == 3) {
#pragma unroll 1
        for (int i = 0; i < iters; ++i) p = xyzz_add(p, q);
    } else {
#pragma unroll 1
        for (int i = 0; i < iters; ++i) p = xyzz_add_par4(p, q);
    }
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        xyzz_store(out, p);
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

int main() {
    uint8_t h[2 * XYZZ_BYTES] = {0};
    // not a curve point: latency only (keep zz != 0 so nothing short-circuits)
    for (int i = 0; i < 2 * XYZZ_BYTES; ++i) h[i] = (uint8_t)(i * 37 + 11) & 0x0f;
    uint8_t *d_in, *d_out;
    uint64_t* d_clk;
    hipMalloc(&d_in, sizeof(h));
    hipMalloc(&d_out, XYZZ_BYTES);
    hipMalloc(&d_clk, 16);
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[5] = {"fq_mul", "xyzz_double (1 lane)", "xyzz_double_par4", "xyzz_add (1 lane)", "xyzz_add_par4"};
    const int iters[5] = {8192, 1024, 1024, 1024, 1024};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 5; ++m) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, m, iters[m], d_in, d_out, d_clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            uint64_t clk[2];
            hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
            if (rep)
                printf("%-22s %8.3f us/op   %8.0f shader-clk/op   shader clock ~%.2f GHz (wall_clock64 at 100 MHz)\n", names[m],
                       ms * 1e3 / iters[m], (double)clk[0] / iters[m], (double)clk[0] / ((double)clk[1] * 10.0) );
        }
    return 0;
}
