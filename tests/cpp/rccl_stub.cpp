// TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that runs every "rank" of a communicator on ONE device, so that the
// single-process, N-context branch of csrc/comm.inc (h2agg_comm_create + h2agg_allgather_add_points with nctx == world:
// ncclCommInitAll, ncclGroupStart / ncclAllGather per context / ncclGroupEnd) can execute on a one-GPU box.  The all-gather is
// plain device-to-device copies at ncclGroupEnd.  Built by tests/test_gpu_comm_group.py with the soname librccl.so.1 and
// loaded into the driver's process BEFORE libh2agg.so looks for RCCL (its dlopen(RTLD_NOLOAD) then finds this one).
// Only the entry points csrc/comm.inc resolves.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>

extern "C" {
typedef struct StubComm { int rank, size, id; }* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // 0 = ncclSuccess
typedef int ncclDataType_t;

struct Pending { const void* send; void* recv; size_t bytes; StubComm* comm; hipStream_t stream; };
static std::vector<Pending> g_pending;
static int g_group_depth = 0, g_next_id = 1, g_allgathers = 0, g_groups = 0;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5a, sizeof *id); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId, int rank) {
    if (n != 1) return 5;   // one process here: only world 1 makes sense in rank mode
    *c = new StubComm{rank, n, g_next_id++};
    return 0;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
    const int id = g_next_id++;
    for (int i = 0; i < n; ++i) comms[i] = new StubComm{i, n, id};
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return 0; }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->size; return 0; }
static ncclResult_t flush() {
    // every rank's recv = [rank 0's send | rank 1's send | ...]; all contexts live on one device
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    for (const Pending& dst : g_pending)
        for (const Pending& src : g_pending)
            if (src.comm->id == dst.comm->id &&
                hipMemcpy((char*)dst.recv + src.bytes * src.comm->rank, src.send, src.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                return 1;
    for (const Pending& p : g_pending)
        if ((int)g_pending.size() < p.comm->size) return 3;   // a rank did not call: a real RCCL would hang
    g_pending.clear();
    return 0;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
    ++g_allgathers;
    g_pending.push_back({send, recv, count, c, s});
    return g_group_depth ? 0 : flush();
}
ncclResult_t ncclGroupStart() { ++g_group_depth; ++g_groups; return 0; }
ncclResult_t ncclGroupEnd() { return --g_group_depth == 0 ? flush() : 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : r == 3 ? "stub: a rank of the communicator did not call" : "stub: failure"; }
// for the driver's assertions
int rccl_stub_allgathers() { return g_allgathers; }
int rccl_stub_groups() { return g_groups; }
}
