#include <cstdio>
#include <chrono>
#include "poseidon_sponge_host.hpp"
using namespace h2agg::poseidon_host;
int main() {
    uint64_t w0[4]={3,5,7,9}; HFr a = one(), b = from_words(w0);
    auto t0 = std::chrono::steady_clock::now();
    const int N = 2000000;
    for (int i = 0; i < N; ++i) a = fmul(a, b);   // dependent chain (latency)
    auto t1 = std::chrono::steady_clock::now();
    HFr x[8]; for (int k = 0; k < 8; ++k) { uint64_t w1[4]={(uint64_t)k+2,5,7,9}; x[k] = from_words(w1); }
    for (int i = 0; i < N / 8; ++i) for (int k = 0; k < 8; ++k) x[k] = fmul(x[k], b);   // 8 independent chains (throughput)
    auto t2 = std::chrono::steady_clock::now();
    HFr o = mul(a, b);
    for (int i = 0; i < N; ++i) o = mul(o, b);
    auto t3 = std::chrono::steady_clock::now();
    uint64_t s = a.l[0] ^ o.l[0]; for (int k = 0; k < 8; ++k) s ^= x[k].l[0];
    printf("fmul latency %.1f ns, throughput %.1f ns, old mul %.1f ns  (%llx)\n",
      std::chrono::duration<double, std::nano>(t1 - t0).count() / N, std::chrono::duration<double, std::nano>(t2 - t1).count() / N,
      std::chrono::duration<double, std::nano>(t3 - t2).count() / N, (unsigned long long)s);
}
