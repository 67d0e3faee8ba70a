// Host-to-device bandwidth from page-locked memory: one stream vs several streams with the buffer cut into pieces
// (does one SDMA engine saturate the link, or do two copies in flight go faster?)  Sizes of the 2^20-point MSM inputs: 96 MiB.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_h2d.hip -o tools/ubench_h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t total = 96u << 20;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, total) != hipSuccess || hipMalloc(&d, total) != hipSuccess) return 1;
    memset(h, 1, total);
    hipStream_t st[8];
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int streams : {1, 2, 4, 8})
        for (size_t piece : {total, total / 2, total / 4, total / 8, total / 16}) {
            if (total / piece < (size_t)streams) continue;
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipDeviceSynchronize();
                const double t0 = now_ms();
                size_t k = 0;
                for (size_t off = 0; off < total; off += piece, ++k)
                    hipMemcpyAsync((char*)d + off, (char*)h + off, piece, hipMemcpyHostToDevice, st[k % streams]);
                for (int s = 0; s < streams; ++s) hipStreamSynchronize(st[s]);
                const double dt = now_ms() - t0;
                if (dt < best) best = dt;
            }
            printf("streams %d  piece %3zu MiB  %.3f ms  %.1f GB/s\n", streams, piece >> 20, best, total / best / 1e6);
        }
    return 0;
}
