// Development harness for csrc/lp_kernels.hpp: the limb-parallel product and doubling against fp.hpp / g1.hpp on random inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I halo2-snark-aggregator_amd/csrc tools/lp_test.hip -o /tmp/lp_test && /tmp/lp_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#include "msm_kernels.hpp"
#include "lp_kernels.hpp"
using namespace h2agg;

__global__ void __launch_bounds__(64) k_swap_probe(uint32_t* out) {
    const uint32_t v = threadIdx.x;
    uint32_t r[4];
    lp_all_rows(v, r);
    for (int q = 0; q < 4; ++q) out[q * 64 + threadIdx.x] = r[q];
}
// one wave per test case: in[case] = two field elements (2 x 9 limbs), out: single-lane product and limb-parallel product
__global__ void __launch_bounds__(64) k_mul_test(const uint32_t* in, uint32_t* out) {
    const LpConst k = lp_const();
    const uint32_t* a = in + 18 * blockIdx.x;
    Fq A, B;
    for (int i = 0; i < NL; ++i) { A.l[i] = a[i]; B.l[i] = a[9 + i]; }
    const Fq want = fp_canonical<FqParams>(FQ_MUL(A, B));
    const Fq want_s = fp_canonical<FqParams>(FQ_SUB(8, A, B));   // (a - b + 9 p and a - b + 8 p: equal mod p)
    const uint32_t la = k.j < NL ? a[k.j] : 0u, lb = k.j < NL ? a[9 + k.j] : 0u;
    const uint32_t got = lp_mul(la, lb, k), got_s = lp_sub<9>(la, lb, k);
    // gather the row-0 limbs into one element through memory, canonicalise on lane 0
    __shared__ uint32_t sm[4][64];
    sm[0][threadIdx.x] = got;
    sm[1][threadIdx.x] = got_s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int row = 0; row < 4; ++row) {
            Fq g, gs;
            for (int i = 0; i < NL; ++i) { g.l[i] = sm[0][16 * row + i]; gs.l[i] = sm[1][16 * row + i]; }
            // nearly tight limbs: normalise before the canonicalising product
            int32_t x[NL];
            for (int i = 0; i < NL; ++i) x[i] = (int32_t)g.l[i];
            g = fp_normalize<FqParams>(x);
            for (int i = 0; i < NL; ++i) x[i] = (int32_t)gs.l[i];
            gs = fp_normalize<FqParams>(x);
            const Fq c = fp_canonical<FqParams>(g), cs = fp_canonical<FqParams>(gs);
            uint32_t bad = 0;
            for (int i = 0; i < NL; ++i) bad |= (c.l[i] ^ want.l[i]) | (cs.l[i] ^ want_s.l[i]);
            out[4 * blockIdx.x + row] = bad;
        }
    }
}
__global__ void __launch_bounds__(64) k_dbl_test(const uint8_t* recs, uint8_t* out_ref, uint8_t* out_lp, int ndbl) {
    const LpConst k = lp_const();
    const uint8_t* rec = recs + XYZZ_BYTES * (size_t)blockIdx.x;
    G1XYZZ p = xyzz_load(rec);
    LpPoint q = lp_load(rec, k);
    for (int i = 0; i < ndbl; ++i) {
        if (threadIdx.x == 0) p = xyzz_double(p);
        q = lp_double(q, k);
    }
    __shared__ uint32_t sm[4][64];
    sm[0][threadIdx.x] = q.x; sm[1][threadIdx.x] = q.y; sm[2][threadIdx.x] = q.zz; sm[3][threadIdx.x] = q.zzz;
    __syncthreads();
    if (threadIdx.x == 0) {
        G1XYZZ g;
        Fq* cs[4] = {&g.x, &g.y, &g.zz, &g.zzz};
        for (int c = 0; c < 4; ++c) {
            int32_t x[NL];
            for (int i = 0; i < NL; ++i) x[i] = (int32_t)sm[c][i];
            *cs[c] = fp_normalize<FqParams>(x);
        }
        // compare as canonical Jacobian encodings (x = X ZZ^2 ...): both through the same conversion
        jac_store_canonical(out_ref + 96 * (size_t)blockIdx.x, jac_from_xyzz(p));
        // bounds of the limb-parallel point are larger (< 40 p): bring the coordinates under 2 p first
        g.x = FQ_MUL(g.x, Fq::one()); g.y = FQ_MUL(g.y, Fq::one()); g.zz = FQ_MUL(g.zz, Fq::one()); g.zzz = FQ_MUL(g.zzz, Fq::one());
        jac_store_canonical(out_lp + 96 * (size_t)blockIdx.x, jac_from_xyzz(g));
    }
}
// pairs (a, b) of XYZZ records: a + b by xyzz_add (lane 0) and by lp_add_points
__global__ void __launch_bounds__(64) k_add_test(const uint8_t* recs, uint8_t* out_ref, uint8_t* out_lp) {
    __shared__ uint32_t sm[4 * 64];
    const LpConst k = lp_const();
    const uint8_t* ra = recs + XYZZ_BYTES * (size_t)(2 * blockIdx.x), *rb = ra + XYZZ_BYTES;
    const G1XYZZ want = xyzz_add(xyzz_load(ra), xyzz_load(rb));
    const LpPoint got = lp_add_points(lp_load(ra, k), lp_load(rb, k), k, sm);
    G1XYZZ g = lp_to_single(got, sm);
    if (threadIdx.x == 0) {
        if (!g.is_identity()) { g.x = FQ_MUL(g.x, Fq::one()); g.y = FQ_MUL(g.y, Fq::one()); }
        jac_store_canonical(out_ref + 96 * (size_t)blockIdx.x, jac_from_xyzz(want));
        jac_store_canonical(out_lp + 96 * (size_t)blockIdx.x, jac_from_xyzz(g));
    }
}
// records k * (1, 2) for k = 1 .. n made on the device (running additions on lane 0)
__global__ void k_make_multiples(uint8_t* recs, int n) {
    G1Affine g1;
    g1.x = Fq::one();
    g1.y = fp_dbl<FqParams>(Fq::one());
    G1XYZZ acc = G1XYZZ::identity();
    for (int i = 0; i < n; ++i) {
        xyzz_add_affine(acc, g1);
        xyzz_store(recs + XYZZ_BYTES * (size_t)i, acc);
    }
}
__global__ void __launch_bounds__(64) k_add_chain(const uint8_t* recs, uint8_t* out, int n) {
    __shared__ uint32_t sm[4 * 64];
    const LpConst k = lp_const();
    LpPoint acc = lp_load(recs, k);
    const LpPoint q = lp_load(recs + XYZZ_BYTES * 5, k);
    for (int i = 0; i < n; ++i) acc = lp_add_points(acc, q, k, sm);
    G1XYZZ g = lp_to_single(acc, sm);
    if (threadIdx.x == 0) xyzz_store(out, g);
}
__global__ void __launch_bounds__(64) k_dbl_chain(const uint8_t* recs, uint8_t* out, int n) {
    __shared__ uint32_t sm[4 * 64];
    const LpConst k = lp_const();
    LpPoint acc = lp_load(recs, k);
    for (int i = 0; i < n; ++i) acc = lp_double(acc, k);
    G1XYZZ g = lp_to_single(acc, sm);
    if (threadIdx.x == 0) xyzz_store(out, g);
}
__global__ void __launch_bounds__(64) k_mul_chain(uint32_t* out, int n) {
    const LpConst k = lp_const();
    uint32_t a = k.j < NL ? FqParams::R2[k.j < NL ? k.j : 0] : 0u, b = a;
    for (int i = 0; i < n; ++i) a = lp_mul(a, b, k);
    out[threadIdx.x] = a;
}
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 256);
    hipLaunchKernelGGL(k_swap_probe, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < 4; ++q) for (int l = 0; l < 64; ++l) if (h[q * 64 + l] != (uint32_t)(16 * q + (l & 15))) ++bad;
    printf("lp_all_rows: %s", bad ? "WRONG semantics:" : "ok\n");
    if (bad) { for (int q = 0; q < 4; ++q) { printf("\n r[%d]:", q); for (int l = 0; l < 64; l += 5) printf(" %u", h[q * 64 + l]); } printf("\n"); }
    // products
    const int N = 4096;
    std::vector<uint32_t> in(18 * N);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& w : in) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s & M29; }
    for (int i = 0; i < N; ++i) { in[18 * i + 8] &= 0x3fffff; in[18 * i + 17] &= 0x3fffff; }   // < 2^254-ish: a few p
    for (int i = 0; i < 9; ++i) { in[i] = 0; in[9 + i] = (i == 0); }                            // 0 * 1
    uint32_t *din, *dout; hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 4 * N * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mul_test, dim3(N), dim3(64), 0, 0, din, dout);
    std::vector<uint32_t> ho(4 * N); hipMemcpy(ho.data(), dout, 4 * N * 4, hipMemcpyDeviceToHost);
    int nb = 0; for (auto v : ho) nb += v != 0;
    printf("lp_mul / lp_sub vs fp_mul / fp_sub on %d cases x 4 rows: %d wrong\n", N, nb);
    // doublings: points k * G made with the group law itself
    const int NP = 256;
    std::vector<uint8_t> recs(XYZZ_BYTES * NP, 0);
    // (1, 2) in Montgomery form repeated, then doubled i times on the device by the reference path itself: use k_dbl_test with ndbl = 0 .. to fill
    uint32_t one[9], two[9];
    for (int i = 0; i < 9; ++i) one[i] = FqParams::R1[i];
    { uint32_t c = 0; for (int i = 0; i < 9; ++i) { uint32_t t = one[i] * 2 + c; two[i] = i < 8 ? t & M29 : t; c = i < 8 ? t >> 29 : 0; } }
    for (int p = 0; p < NP; ++p) {
        uint32_t* w = (uint32_t*)(recs.data() + XYZZ_BYTES * p);
        for (int i = 0; i < 9; ++i) { w[i] = one[i]; w[9 + i] = two[i]; w[18 + i] = one[i]; w[27 + i] = one[i]; }
    }
    uint8_t *drec, *dref, *dlp; hipMalloc(&drec, recs.size()); hipMalloc(&dref, 96 * NP); hipMalloc(&dlp, 96 * NP);
    hipMemcpy(drec, recs.data(), recs.size(), hipMemcpyHostToDevice);
    int wrong = 0;
    for (int nd : {1, 2, 3, 17, 64}) {
        hipLaunchKernelGGL(k_dbl_test, dim3(NP), dim3(64), 0, 0, drec, dref, dlp, nd);
        std::vector<uint8_t> a(96 * NP), b(96 * NP);
        hipMemcpy(a.data(), dref, a.size(), hipMemcpyDeviceToHost); hipMemcpy(b.data(), dlp, b.size(), hipMemcpyDeviceToHost);
        // Jacobian encodings differ by Z: compare affine x = X / Z^2 by cross-multiplication is overkill here: both conversions use Z = ZZZ,
        // and the two paths compute the SAME XYZZ formulas, so the coordinates agree mod p -> identical canonical bytes
        int w = memcmp(a.data(), b.data(), a.size()) != 0;
        printf("lp_double x %d vs xyzz_double: %s\n", nd, w ? "DIFFERENT" : "equal");
        wrong += w;
    }
    // additions: multiples of G in XYZZ form; pairs include P + P, P + (-P), identity operands
    const int NM = 64;
    uint8_t* dm; hipMalloc(&dm, XYZZ_BYTES * NM);
    hipLaunchKernelGGL(k_make_multiples, dim3(1), dim3(1), 0, 0, dm, NM);
    std::vector<uint8_t> mult(XYZZ_BYTES * NM); hipMemcpy(mult.data(), dm, mult.size(), hipMemcpyDeviceToHost);
    std::vector<uint8_t> pairs;
    auto push = [&](const uint8_t* r) { pairs.insert(pairs.end(), r, r + XYZZ_BYTES); };
    std::vector<uint8_t> ident(XYZZ_BYTES, 0);
    { uint32_t* w = (uint32_t*)ident.data(); for (int i = 0; i < 9; ++i) w[9 + i] = FqParams::R1[i]; }   // (0, 1, 0, 0)
    int npairs = 0;
    for (int i = 0; i < NM; ++i) for (int j : {0, 1, 5, 31, i}) { push(&mult[XYZZ_BYTES * i]); push(&mult[XYZZ_BYTES * (j % NM)]); ++npairs; }
    for (int i = 0; i < 8; ++i) {   // P + (-P): negate y = 2 p - y limbwise is not tight; use 4p - y through the record of 2 * i-th multiple: build -P as (x, -y)
        std::vector<uint8_t> neg(&mult[XYZZ_BYTES * i], &mult[XYZZ_BYTES * i] + XYZZ_BYTES);
        uint32_t* w = (uint32_t*)neg.data();
        // y <- 8 p - y with a borrow-propagating subtraction on 29-bit limbs
        uint32_t kp[9]; { uint64_t c = 0; for (int q = 0; q < 9; ++q) { uint64_t t = (uint64_t)FqParams::MOD[q] * 8 + c; kp[q] = q < 8 ? (uint32_t)(t & M29) : (uint32_t)t; c = t >> 29; } }
        int64_t br = 0;
        for (int q = 0; q < 9; ++q) { int64_t t = (int64_t)kp[q] - w[9 + q] + br; if (q < 8) { w[9 + q] = (uint32_t)(t & M29); br = t >> 29; } else w[9 + q] = (uint32_t)t; }
        push(&mult[XYZZ_BYTES * i]); push(neg.data()); ++npairs;
        push(ident.data()); push(&mult[XYZZ_BYTES * i]); ++npairs;
        push(&mult[XYZZ_BYTES * i]); push(ident.data()); ++npairs;
    }
    uint8_t *dp, *dr2, *dl2; hipMalloc(&dp, pairs.size()); hipMalloc(&dr2, 96 * npairs); hipMalloc(&dl2, 96 * npairs);
    hipMemcpy(dp, pairs.data(), pairs.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_add_test, dim3(npairs), dim3(64), 0, 0, dp, dr2, dl2);
    std::vector<uint8_t> ra(96 * npairs), rb(96 * npairs);
    hipMemcpy(ra.data(), dr2, ra.size(), hipMemcpyDeviceToHost); hipMemcpy(rb.data(), dl2, rb.size(), hipMemcpyDeviceToHost);
    int addbad = 0;
    for (int i = 0; i < npairs; ++i) {
        // the two paths may return different Jacobian representatives only if they took different formulas (the exceptional route
        // of lp_add_points IS xyzz_add): compare affine by cross-multiplication on the host is avoided by comparing bytes first
        if (memcmp(&ra[96 * i], &rb[96 * i], 96) != 0) ++addbad;
    }
    printf("lp_add_points vs xyzz_add on %d pairs (incl. P + P, P - P, identity operands): %d different\n", npairs, addbad);
    // the Horner kernel: W window sums = multiples of G, c doublings per window
    int hbad = 0;
    for (int cW : {16 * 100 + 8, 16 * 100 + 16, 13 * 100 + 10, 8 * 100 + 16}) {
        const int c = cW / 100, W = cW % 100;
        uint8_t *dj1, *dj2, *dx1, *dx2; hipMalloc(&dj1, 96); hipMalloc(&dj2, 96); hipMalloc(&dx1, XYZZ_BYTES); hipMalloc(&dx2, XYZZ_BYTES);
        hipLaunchKernelGGL(k_msm_final, dim3(1), dim3(64), 0, 0, dm + XYZZ_BYTES * 3, c, W, dx1, dj1);
        hipLaunchKernelGGL(k_msm_final_lp, dim3(1), dim3(64), 0, 0, dm + XYZZ_BYTES * 3, c, W, dx2, dj2);
        uint8_t j1[96], j2[96];
        hipMemcpy(j1, dj1, 96, hipMemcpyDeviceToHost); hipMemcpy(j2, dj2, 96, hipMemcpyDeviceToHost);
        // time both
        hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_msm_final, dim3(1), dim3(64), 0, 0, dm + XYZZ_BYTES * 3, c, W, dx1, dj1);
        hipEventRecord(e1);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_msm_final_lp, dim3(1), dim3(64), 0, 0, dm + XYZZ_BYTES * 3, c, W, dx2, dj2);
        hipEventRecord(e2); hipEventSynchronize(e2);
        float t1, t2; hipEventElapsedTime(&t1, e0, e1); hipEventElapsedTime(&t2, e1, e2);
        const int same = memcmp(j1, j2, 96) == 0;
        printf("Horner c = %d, W = %d: k_msm_final %.1f us, k_msm_final_lp %.1f us, results %s\n", c, W, t1 * 200, t2 * 200, same ? "equal" : "DIFFERENT");
        hbad += !same;
    }
    {   // latencies of the primitives on a lone wave
        uint8_t* dx; hipMalloc(&dx, XYZZ_BYTES); uint32_t* dw; hipMalloc(&dw, 256);
        hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
        hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(64), 0, 0, dm, dx, 8);
        hipEventRecord(e[0]);
        hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(64), 0, 0, dm, dx, 200);
        hipEventRecord(e[1]);
        hipLaunchKernelGGL(k_dbl_chain, dim3(1), dim3(64), 0, 0, dm, dx, 200);
        hipEventRecord(e[2]);
        hipLaunchKernelGGL(k_mul_chain, dim3(1), dim3(64), 0, 0, dw, 1000);
        hipEventRecord(e[3]); hipEventSynchronize(e[3]);
        float ta, td, tm; hipEventElapsedTime(&ta, e[0], e[1]); hipEventElapsedTime(&td, e[1], e[2]); hipEventElapsedTime(&tm, e[2], e[3]);
        printf("lone wave: lp_add_points %.2f us, lp_double %.2f us, lp_mul %.3f us each\n", ta * 5, td * 5, tm);
    }
    return (bad || nb || wrong || addbad || hbad) ? 1 : 0;
}
