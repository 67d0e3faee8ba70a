// TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that runs every "rank" of a communicator on ONE device, so that the
// single-process, N-context branch of csrc/comm.inc (h2agg_comm_create + h2agg_allgather_add_points with nctx == world:
// ncclCommInitAll, ncclGroupStart / ncclAllGather per context / ncclGroupEnd) can execute on a one-GPU box.  The all-gather is
// plain device-to-device copies at ncclGroupEnd.  Built by tests/test_gpu_comm_group.py with the soname librccl.so.1 and
// loaded into the driver's process BEFORE libh2agg.so looks for RCCL (its dlopen(RTLD_NOLOAD) then finds this one).
// Only the entry points csrc/comm.inc resolves.  Not part of the product.
//
// Second mode (tests/test_gpu_sharded.py): ranks as THREADS of one process, one context each, communicators made with
// ncclCommInitRank(n > 1) from one unique id — what one process per GPU does, minus the processes.  ncclAllGather then is a
// rendezvous: a rank blocks until every rank of its communicator has called, the last arrival copies, all return; a rank that
// never calls makes the others fail after a timeout instead of hanging the test (a real RCCL would hang).
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

extern "C" {
typedef struct StubComm { int rank, size, id; bool threaded; }* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // 0 = ncclSuccess
typedef int ncclDataType_t;

struct Pending { const void* send; void* recv; size_t bytes; StubComm* comm; hipStream_t stream; };
static std::vector<Pending> g_pending;
static int g_group_depth = 0, g_next_id = 1, g_allgathers = 0, g_groups = 0;

static std::mutex g_mu;
static std::condition_variable g_cv;
struct Rendezvous { std::vector<Pending> p; unsigned generation = 0; };
static std::map<int, Rendezvous> g_rdv;   // per communicator id (threads-as-ranks mode)

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0x5a, sizeof *id);
    const int key = 1000000 + g_next_id++;   // every id names its own group of ranks
    memcpy(id->internal, &key, sizeof key);
    return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId id, int rank) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (n < 1 || rank < 0 || rank >= n) return 5;
    int key;
    memcpy(&key, id.internal, sizeof key);
    *c = new StubComm{rank, n, n == 1 ? g_next_id++ : key, n > 1};
    return 0;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
    const int id = g_next_id++;
    for (int i = 0; i < n; ++i) comms[i] = new StubComm{i, n, id, false};
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
    if (c->threaded) {   // ranks are threads: wait for the others
        if (hipStreamSynchronize(s) != hipSuccess) return 1;   // (the send buffer is ready; the stub copies synchronously)
        std::unique_lock<std::mutex> lk(g_mu);
        ++g_allgathers;
        Rendezvous& r = g_rdv[c->id];
        const unsigned gen = r.generation;
        r.p.push_back({send, recv, count, c, s});
        if ((int)r.p.size() == c->size) {
            ncclResult_t rc = 0;
            for (const Pending& dst : r.p)
                for (const Pending& src : r.p)
                    if (src.bytes != dst.bytes ||
                        hipMemcpy((char*)dst.recv + src.bytes * src.comm->rank, src.send, src.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                        rc = 1;
            r.p.clear();
            ++r.generation;
            g_cv.notify_all();
            return rc;
        }
        if (!g_cv.wait_for(lk, std::chrono::seconds(120), [&] { return r.generation != gen; })) return 3;   // a rank never came
        return 0;
    }
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
