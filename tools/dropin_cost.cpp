// What the DROP-IN MockEccChip::multi_exp costs end to end at 2^20 pairs — a C++ stand-in for rust-shim's
// multi_exp_with_point_list (no Rust toolchain in the image): the calling thread marshals points (x, y, z out of Montgomery
// form) and scalars into page-locked buffers from h2agg_host_alloc and calls h2agg_g1_msm_jac, WHILE worker threads produce
// `ctx.point_list = points.map(|x| format!("{:?}", x))` (mock/arith/ecc.rs:112-116: three coordinates as 0x + 64 hex digits,
// one heap string per point), one contiguous chunk each.  Printed: the phases alone and the overlapped total.
//   g++ -O2 -std=c++17 -pthread -I include tools/dropin_cost.cpp -o /tmp/dropin_cost -L halo2-snark-aggregator_amd -lh2agg \
//       -Wl,-rpath,$PWD/halo2-snark-aggregator_amd -Wl,-rpath,/opt/rocm/lib && /tmp/dropin_cost [threads]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "h2agg.h"
typedef unsigned __int128 u128;
static const uint64_t MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t INV = 0x87d20782e4866389ull;
static void from_mont(const uint64_t a[4], uint64_t out[4]) {   // a * 1 / 2^256 mod p (what to_repr() / Debug do first)
    uint64_t t[5] = {a[0], a[1], a[2], a[3], 0};
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[0] * INV;
        u128 c = (u128)m * MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = (uint64_t)(c >> 64);
    }
    for (int i = 0; i < 4; ++i) out[i] = t[i];
}
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char** argv) {
    const size_t n = 1 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned workers = argc > 1 ? (unsigned)atoi(argv[1]) : (hw > 1 ? hw - 1 : 1);
    h2agg_ctx* c = nullptr;
    if (h2agg_create(0, &c) != 0) { fprintf(stderr, "h2agg_create failed\n"); return 1; }
    // the caller's data: 2^20 points k_i * G as the library returns them (canonical Jacobian), held in "Montgomery form" the
    // way halo2curves holds coordinates (the stand-in only needs the conversion's cost, so the bytes are reused as limbs)
    std::vector<uint8_t> g(64 * n, 0), sc(32 * n), pts(96 * n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        g[64 * i] = 1; g[64 * i + 32] = 2;
        for (int k = 0; k < 4; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(&sc[32 * i + 8 * k], &s, 8); }
        sc[32 * i + 31] &= 0x1f;
    }
    if (h2agg_g1_batch_scalar_mul(c, g.data(), sc.data(), n, pts.data()) != 0) { fprintf(stderr, "setup: %s\n", h2agg_last_error(c)); return 1; }
    uint8_t *hp = nullptr, *hs = nullptr;
    if (h2agg_host_alloc(c, 96 * n, (void**)&hp) != 0 || h2agg_host_alloc(c, 32 * n, (void**)&hs) != 0) return 1;
    uint8_t out[96];
    auto marshal = [&] {   // 3 coordinates + 1 scalar out of Montgomery form per pair, straight into the page-locked buffers
        const uint64_t* src = (const uint64_t*)pts.data();
        const uint64_t* ssrc = (const uint64_t*)sc.data();
        for (size_t i = 0; i < n; ++i) {
            uint64_t w[4];
            for (int k = 0; k < 3; ++k) { from_mont(src + 12 * i + 4 * k, w); (void)w; }
            from_mont(ssrc + 4 * i, w);
            memcpy(hp + 96 * i, pts.data() + 96 * i, 96);   // (the canonical bytes themselves: the MSM must see valid points)
            memcpy(hs + 32 * i, sc.data() + 32 * i, 32);
        }
    };
    static const char* hexd = "0123456789abcdef";
    auto format_chunk = [&](size_t lo, size_t hi, std::vector<std::string>* list) {
        const uint64_t* src = (const uint64_t*)pts.data();
        list->reserve(hi - lo);
        for (size_t i = lo; i < hi; ++i) {
            std::string o;
            o.reserve(224);
            o += "(";
            for (int k = 0; k < 3; ++k) {
                uint64_t w[4];
                from_mont(src + 12 * i + 4 * k, w);
                o += "0x";
                for (int q = 3; q >= 0; --q)
                    for (int b = 60; b >= 0; b -= 4) o += hexd[(w[q] >> b) & 15];
                o += k < 2 ? ", " : ")";
            }
            list->push_back(std::move(o));
        }
    };
    for (int r = 0; r < 2; ++r) { marshal(); h2agg_g1_msm_jac(c, hp, hs, n, out); }   // warm-up
    auto t0 = std::chrono::steady_clock::now();
    marshal();
    const double t_marshal = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (h2agg_g1_msm_jac(c, hp, hs, n, out) != 0) { fprintf(stderr, "msm: %s\n", h2agg_last_error(c)); return 1; }
    const double t_gpu = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    { std::vector<std::string> one; format_chunk(0, n, &one); }
    const double t_fmt1 = ms_since(t0);
    // overlapped: workers format, this thread marshals + calls the GPU
    std::vector<std::vector<std::string>> parts(workers);
    t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        const size_t chunk = (n + workers - 1) / workers;
        for (unsigned w = 0; w < workers; ++w) {
            const size_t lo = w * chunk, hi = lo + chunk < n ? lo + chunk : n;
            if (lo < hi) th.emplace_back(format_chunk, lo, hi, &parts[w]);
        }
        marshal();
        h2agg_g1_msm_jac(c, hp, hs, n, out);
        for (auto& t : th) t.join();
    }
    std::vector<std::string> list;
    list.reserve(n);
    for (auto& p : parts) for (auto& x : p) list.push_back(std::move(x));
    const double t_all = ms_since(t0);
    printf("2^20 pairs: marshal %.1f ms (1 thread) | h2agg_g1_msm_jac from page-locked buffers %.2f ms | point_list %.1f ms on 1 thread\n", t_marshal, t_gpu, t_fmt1);
    printf("drop-in multi_exp, point_list on %u worker threads under the marshalling + GPU call: %.1f ms end to end = %.1f M points/s"
           "   (serial, as the reference orders it: %.1f ms = %.2f M points/s)\n",
           workers, t_all, n / t_all / 1e3, t_marshal + t_gpu + t_fmt1, n / (t_marshal + t_gpu + t_fmt1) / 1e3);
    h2agg_host_free(c, hp); h2agg_host_free(c, hs); h2agg_destroy(c);
    return list.size() == n ? 0 : 1;
}
