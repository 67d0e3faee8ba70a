// Drives the single-process, N-context branch of the exchange (h2agg_comm_create + h2agg_allgather_add_points with
// nctx == world; halo2-snark-aggregator_amd/csrc/comm.inc) through the C ABI on ONE GPU, with tests/cpp/rccl_stub.cpp standing
// in for RCCL.  usage: comm_group_driver <path to the stub librccl.so.1> <ranks>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "h2agg.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, ctxs[0] ? h2agg_last_error(ctxs[0]) : "-"); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: comm_group_driver <stub librccl.so.1> <ranks>\n"); return 2; }
    void* stub = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!stub) { fprintf(stderr, "dlopen stub: %s\n", dlerror()); return 2; }
    const int world = atoi(argv[2]), npts = 3;
    std::vector<int> devs(world, 0);                 // every "rank" on device 0
    std::vector<h2agg_ctx*> ctxs(world, nullptr);
    CHECK(h2agg_comm_create(devs.data(), world, ctxs.data()));
    for (int r = 0; r < world; ++r)
        if (h2agg_comm_size(ctxs[r]) != world || h2agg_comm_rank(ctxs[r]) != r) { fprintf(stderr, "rank/size of context %d\n", r); return 1; }
    // partial[r][k] = (100 r + k + 1) * G as canonical Jacobian; the fold of point k is (sum_r (100 r + k + 1)) * G
    uint8_t g[64] = {0};
    g[0] = 1; g[32] = 2;
    std::vector<uint8_t> bases(64 * world * npts), scal(32 * world * npts, 0), partial(96 * world * npts);
    for (int i = 0; i < world * npts; ++i) {
        memcpy(bases.data() + 64 * i, g, 64);
        const unsigned v = 100u * (i / npts) + (i % npts) + 1u;
        scal[32 * i] = v & 0xff; scal[32 * i + 1] = v >> 8;
    }
    CHECK(h2agg_g1_batch_scalar_mul(ctxs[0], bases.data(), scal.data(), world * npts, partial.data()));
    std::vector<uint8_t> want_s(32 * npts, 0), want_j(96 * npts), want(64 * npts), got(64 * npts);
    for (int k = 0; k < npts; ++k) {
        unsigned v = 0;
        for (int r = 0; r < world; ++r) v += 100u * r + k + 1u;
        want_s[32 * k] = v & 0xff; want_s[32 * k + 1] = (v >> 8) & 0xff; want_s[32 * k + 2] = v >> 16;
    }
    CHECK(h2agg_g1_batch_scalar_mul(ctxs[0], bases.data(), want_s.data(), npts, want_j.data()));
    CHECK(h2agg_g1_batch_to_affine(ctxs[0], want_j.data(), npts, want.data()));
    CHECK(h2agg_allgather_add_points(ctxs.data(), world, partial.data(), npts, got.data()));
    if (memcmp(got.data(), want.data(), want.size()) != 0) { fprintf(stderr, "folded points differ\n"); return 1; }
    // passing only SOME of the communicator's contexts is refused (a real RCCL would hang on the missing ranks)
    if (world > 2 && h2agg_allgather_add_points(ctxs.data(), 2, partial.data(), npts, got.data()) == 0) { fprintf(stderr, "partial group accepted\n"); return 1; }
    int (*n_ag)() = (int (*)())dlsym(stub, "rccl_stub_allgathers");
    int (*n_gr)() = (int (*)())dlsym(stub, "rccl_stub_groups");
    if (!n_ag || n_ag() != world || n_gr() != (world > 1 ? 1 : 0)) { fprintf(stderr, "stub saw %d all-gathers in %d groups\n", n_ag ? n_ag() : -1, n_gr ? n_gr() : -1); return 1; }
    for (h2agg_ctx* c : ctxs) h2agg_destroy(c);
    printf("comm group ok: %d ranks on one device, %d all-gathers in one group\n", world, world);
    return 0;
}
