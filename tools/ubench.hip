// Instruction-rate and field-op throughput microbenchmarks for gfx950 (decides the limb representation).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I halo2-snark-aggregator_amd/csrc tools/ubench.hip -o gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "g1.hpp"
using namespace h2agg;

#define ITERS 2000
#define DEFK(NAME, DECL, BODY)                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {       \
        DECL;                                                                         \
        for (int it = 0; it < ITERS; ++it) { BODY BODY BODY BODY }                    \
        out[blockIdx.x * 256 + threadIdx.x] = sink;                                   \
    }

// 8 independent chains, one instruction each per BODY
#define U32_DECL uint32_t a0=seed+threadIdx.x,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*9,a5=a0*11,a6=a0*13,a7=a0*17,b=seed|1,sink=0
#define U32_SINK sink = a0^a1^a2^a3^a4^a5^a6^a7;
#define ASM8(INS) asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(b)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(b)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(b)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(b));

__global__ void __launch_bounds__(256) k_add_u32(uint32_t* out, uint32_t seed) { U32_DECL;
  for (int it = 0; it < ITERS; ++it) { ASM8("v_add_u32") ASM8("v_add_u32") ASM8("v_add_u32") ASM8("v_add_u32") } U32_SINK out[blockIdx.x*256+threadIdx.x]=sink; }
__global__ void __launch_bounds__(256) k_mul_lo_u32(uint32_t* out, uint32_t seed) { U32_DECL;
  for (int it = 0; it < ITERS; ++it) { ASM8("v_mul_lo_u32") ASM8("v_mul_lo_u32") ASM8("v_mul_lo_u32") ASM8("v_mul_lo_u32") } U32_SINK out[blockIdx.x*256+threadIdx.x]=sink; }
__global__ void __launch_bounds__(256) k_mul_hi_u32(uint32_t* out, uint32_t seed) { U32_DECL;
  for (int it = 0; it < ITERS; ++it) { ASM8("v_mul_hi_u32") ASM8("v_mul_hi_u32") ASM8("v_mul_hi_u32") ASM8("v_mul_hi_u32") } U32_SINK out[blockIdx.x*256+threadIdx.x]=sink; }
__global__ void __launch_bounds__(256) k_mul_u32_u24(uint32_t* out, uint32_t seed) { U32_DECL;
  for (int it = 0; it < ITERS; ++it) { ASM8("v_mul_u32_u24") ASM8("v_mul_u32_u24") ASM8("v_mul_u32_u24") ASM8("v_mul_u32_u24") } U32_SINK out[blockIdx.x*256+threadIdx.x]=sink; }
__global__ void __launch_bounds__(256) k_mul_hi_u32_u24(uint32_t* out, uint32_t seed) { U32_DECL;
  for (int it = 0; it < ITERS; ++it) { ASM8("v_mul_hi_u32_u24") ASM8("v_mul_hi_u32_u24") ASM8("v_mul_hi_u32_u24") ASM8("v_mul_hi_u32_u24") } U32_SINK out[blockIdx.x*256+threadIdx.x]=sink; }

#define MAD8 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b) : "vcc"); \
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c2) : "v"(a2), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c3) : "v"(a3), "v"(b) : "vcc"); \
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c4) : "v"(a4), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c5) : "v"(a5), "v"(b) : "vcc"); \
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c6) : "v"(a6), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c7) : "v"(a7), "v"(b) : "vcc");
__global__ void __launch_bounds__(256) k_mad_u64_u32(uint32_t* out, uint32_t seed) { U32_DECL;
  uint64_t c0=a0,c1=a1,c2=a2,c3=a3,c4=a4,c5=a5,c6=a6,c7=a7;
  for (int it = 0; it < ITERS; ++it) { MAD8 MAD8 MAD8 MAD8 }
  sink = (uint32_t)(c0^c1^c2^c3^c4^c5^c6^c7) ^ (uint32_t)((c0^c1^c2^c3^c4^c5^c6^c7)>>32); out[blockIdx.x*256+threadIdx.x]=sink; }

#define LSH8 asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c0) : "v"(d)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c1) : "v"(d)); \
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c2) : "v"(d)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c3) : "v"(d)); \
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c4) : "v"(d)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c5) : "v"(d)); \
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c6) : "v"(d)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c7) : "v"(d));
__global__ void __launch_bounds__(256) k_lshl_add_u64(uint32_t* out, uint32_t seed) { U32_DECL;
  uint64_t c0=a0,c1=a1,c2=a2,c3=a3,c4=a4,c5=a5,c6=a6,c7=a7,d=((uint64_t)b<<20)|7;
  for (int it = 0; it < ITERS; ++it) { LSH8 LSH8 LSH8 LSH8 }
  sink = (uint32_t)(c0^c1^c2^c3^c4^c5^c6^c7) ^ (uint32_t)((c0^c1^c2^c3^c4^c5^c6^c7)>>32); out[blockIdx.x*256+threadIdx.x]=sink; }

#define FMA8 asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(x), "v"(y)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c1) : "v"(x), "v"(y)); \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c2) : "v"(x), "v"(y)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c3) : "v"(x), "v"(y)); \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c4) : "v"(x), "v"(y)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c5) : "v"(x), "v"(y)); \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c6) : "v"(x), "v"(y)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c7) : "v"(x), "v"(y));
__global__ void __launch_bounds__(256) k_fma_f64(uint32_t* out, uint32_t seed) {
  double x = 1.0 + 1e-9 * (seed + threadIdx.x), y = 1e-12, c0=1,c1=2,c2=3,c3=4,c4=5,c5=6,c6=7,c7=8;
  for (int it = 0; it < ITERS; ++it) { FMA8 FMA8 FMA8 FMA8 }
  out[blockIdx.x*256+threadIdx.x]=(uint32_t)(c0+c1+c2+c3+c4+c5+c6+c7); }
#define FMAF8 asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(x), "v"(y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(x), "v"(y)); \
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c2) : "v"(x), "v"(y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c3) : "v"(x), "v"(y)); \
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c4) : "v"(x), "v"(y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c5) : "v"(x), "v"(y)); \
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c6) : "v"(x), "v"(y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c7) : "v"(x), "v"(y));
__global__ void __launch_bounds__(256) k_fma_f32(uint32_t* out, uint32_t seed) {
  float x = 1.0f + 1e-6f * (seed + threadIdx.x), y = 1e-7f, c0=1,c1=2,c2=3,c3=4,c4=5,c5=6,c6=7,c7=8;
  for (int it = 0; it < ITERS; ++it) { FMAF8 FMAF8 FMAF8 FMAF8 }
  out[blockIdx.x*256+threadIdx.x]=(uint32_t)(c0+c1+c2+c3+c4+c5+c6+c7); }

// whole-op throughput
__global__ void __launch_bounds__(256) k_fqmul(uint32_t* out, uint32_t seed) {
  Fq a = Fq::one(), b = Fq::r2(); a.l[0] ^= (seed + threadIdx.x) & 0xffff;
  for (int it = 0; it < 256; ++it) { a = fp_mul<FqParams>(a, b); b = fp_mul<FqParams>(b, a); }
  out[blockIdx.x*256+threadIdx.x] = a.l[0] ^ b.l[3]; }
__global__ void __launch_bounds__(256) k_fqsqr(uint32_t* out, uint32_t seed) {
  Fq a = Fq::one(), b = Fq::r2(); a.l[0] ^= (seed + threadIdx.x) & 0xffff;
  for (int it = 0; it < 256; ++it) { a = fp_sqr<FqParams>(a); b = fp_sqr<FqParams>(b); }
  out[blockIdx.x*256+threadIdx.x] = a.l[0] ^ b.l[3]; }
__global__ void __launch_bounds__(256) k_fqadd(uint32_t* out, uint32_t seed) {
  Fq a = Fq::one(), b = Fq::r2(); a.l[0] ^= (seed + threadIdx.x) & 0xffff;
  for (int it = 0; it < 2048; ++it) { a = fp_add<FqParams>(a, b); b = fp_sub<8, FqParams>(b, a); a.l[8] &= 0xffff; b.l[8] &= 0xffff; }
  out[blockIdx.x*256+threadIdx.x] = a.l[0] ^ b.l[3]; }
__global__ void __launch_bounds__(256) k_madd(uint32_t* out, uint32_t seed) {
  G1Affine g; g.x = Fq::one(); g.y = fp_dbl<FqParams>(Fq::one());
  G1XYZZ acc = xyzz_double_affine(g); acc.x.l[0] ^= (seed + threadIdx.x) & 1 ? 0 : 0;
  G1Affine q = g;
  for (int it = 0; it < 64; ++it) { xyzz_add_affine(acc, q); }
  out[blockIdx.x*256+threadIdx.x] = acc.x.l[0] ^ acc.zzz.l[3]; }

template <class K> double run(K k, const char* name, double ops_per_thread, int blocks, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 2u + r); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double total = ops_per_thread * blocks * 256.0;
  double rate = total / (ms * 1e-3);
  // cycles per wave-instruction per SIMD at 2.4 GHz: SIMDs = 1024
  double cyc = (2.4e9 * 1024.0) / (rate / 64.0);
  printf("%-18s %8.3f ms  %10.3e ops/s  %6.2f cyc/wave-inst/SIMD(@2.4GHz)\n", name, ms, rate, cyc);
  return rate; }

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int blocks = p.multiProcessorCount * 8; uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  double per = ITERS * 32.0;
  run(k_add_u32, "v_add_u32", per, blocks, d);
  run(k_mul_lo_u32, "v_mul_lo_u32", per, blocks, d);
  run(k_mul_hi_u32, "v_mul_hi_u32", per, blocks, d);
  run(k_mul_u32_u24, "v_mul_u32_u24", per, blocks, d);
  run(k_mul_hi_u32_u24, "v_mul_hi_u32_u24", per, blocks, d);
  run(k_mad_u64_u32, "v_mad_u64_u32", per, blocks, d);
  run(k_lshl_add_u64, "v_lshl_add_u64", per, blocks, d);
  run(k_fma_f64, "v_fma_f64", per, blocks, d);
  run(k_fma_f32, "v_fma_f32", per, blocks, d);
  run(k_fqmul, "fq_mul", 512.0, blocks, d);
  run(k_fqsqr, "fq_sqr", 512.0, blocks, d);
  run(k_fqadd, "fq_add/sub", 4096.0, blocks, d);
  run(k_madd, "xyzz_madd", 64.0, blocks, d);
  return 0; }
