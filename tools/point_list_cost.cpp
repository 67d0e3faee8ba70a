// What `ctx.point_list = points.iter().map(|x| format!("{:?}", x)).collect()` (mock/arith/ecc.rs:112-116) costs on the host
// at 2^20 points — a C++ stand-in (no Rust toolchain in the image): halo2curves' Debug for G1 prints the three coordinates,
// each as 0x + 64 hex digits (Fq's Debug: the canonical integer, big-endian hex), so a point is ~215 characters in its own
// heap-allocated String.  Montgomery -> canonical conversion of the coordinates (what Fq's Debug does first) is included as
// one 4 x 64-bit Montgomery multiplication by 1 per coordinate.
//   g++ -O2 -std=c++17 tools/point_list_cost.cpp -o /tmp/point_list_cost && /tmp/point_list_cost
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
typedef unsigned __int128 u128;
static const uint64_t MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t INV = 0x87d20782e4866389ull;
static void from_mont(const uint64_t a[4], uint64_t out[4]) {   // a * 1 / 2^256 mod p
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
int main() {
    const size_t n = 1 << 20;
    std::vector<uint64_t> pts(12 * n);
    uint64_t s = 88172645463325252ull;
    for (auto& w : pts) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = s; }
    for (size_t i = 0; i < 3 * n; ++i) pts[4 * i + 3] &= 0x1fffffffffffffffull;
    static const char* hexd = "0123456789abcdef";
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::string> list;
    list.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        std::string o;
        o.reserve(224);
        o += "(";
        for (int c = 0; c < 3; ++c) {
            uint64_t w[4];
            from_mont(&pts[12 * i + 4 * c], w);
            o += "0x";
            for (int k = 3; k >= 0; --k)
                for (int b = 60; b >= 0; b -= 4) o += hexd[(w[k] >> b) & 15];
            o += c < 2 ? ", " : ")";
        }
        list.push_back(std::move(o));
    }
    const auto t1 = std::chrono::steady_clock::now();
    size_t bytes = 0;
    for (const auto& x : list) bytes += x.size();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    printf("point_list of %zu points: %.1f ms on one core (%.0f ns per point, %zu MB of strings)\n", n, ms, ms * 1e6 / n, bytes >> 20);
    return 0;
}
