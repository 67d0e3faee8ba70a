// TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that lets every "rank" of a communicator live on ONE device, so that the
// multi-rank branches of the library (csrc/comm.inc, csrc/verifier.inc shard_allgather) can run on a one-GPU box.  The real
// RCCL refuses two ranks on one device.  Only the entry points csrc/comm.inc resolves.  Not part of the product.
//
// What it keeps of RCCL's contract — the part the library relies on:
//   * ncclAllGather is STREAM-ORDERED on the caller's stream.  It never synchronises a stream or the device when all ranks
//     live in this process: the send buffer is read behind an event recorded on the sender's stream at its call, the copies
//     into a rank's recv buffer are enqueued on THAT rank's stream, and a rank's stream does not run past the collective until
//     every peer has pulled its send buffer (events again).  A caller that reads recv without ordering itself behind its own
//     stream, or that refills send from another stream, gets what it would get from RCCL: garbage.
//   * every rank must call; a rank that never does makes the others fail after a timeout (a real RCCL would hang).
// What it does not keep: the HOST blocks in ncclAllGather until every rank of the communicator has made the call (RCCL
// returns after the enqueue and its kernels do the waiting on the device; a kernel spinning for a peer on the same GPU can
// starve that peer, so the rendezvous is on the host here).  Nothing on the device is waited for by the host.
//
// Three ways to hold ranks, one data path:
//   (1) ncclCommInitAll: all ranks in one thread (h2agg_comm_create + h2agg_allgather_add_points with nctx == world);
//       the all-gather happens at ncclGroupEnd, no rendezvous needed.                      tests/cpp/comm_group_driver.cpp
//   (2) ncclCommInitRank, ranks = THREADS of one process (one context each).               tests/rccl_stub_ranks.py
//   (3) ncclCommInitRank, ranks = PROCESSES sharing the device (what one process per GPU does, on one GPU): peers' send
//       buffers are opened with hipIpcOpenMemHandle.  Events do not cross processes here, so in this mode a rank
//       synchronises ITS OWN stream before it publishes its send buffer and after it has pulled (the copies themselves stay
//       on the caller's stream, so the caller's D2H behind the collective is still ordered by the stream alone).
//                                                                                          tests/rccl_stub_ranks.py procs, bench.py
// (2) and (3) meet in a POSIX shared-memory segment named after the unique id; whether a peer is a thread or a process is
// read off its pid.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

extern "C" {
typedef struct StubComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // 0 = ncclSuccess
typedef int ncclDataType_t;
}

namespace {

constexpr int MAX_RANKS = 64;
constexpr int TIMEOUT_S = 120;

struct Slot {   // what a rank publishes for one all-gather
    int pid;
    uint64_t send, bytes, off;      // send: the pointer in the owner's address space; off: its offset in the IPC allocation
    hipIpcMemHandle_t handle;       // (cross-process peers)
    uint64_t ev_ready, ev_pulled;   // hipEvent_t of the owner (same-process peers)
};
struct Seg {   // one per communicator, in shared memory; zero-filled = initial state
    std::atomic<uint32_t> count, gen, broken;
    Slot slot[MAX_RANKS];
};

struct Opened { hipIpcMemHandle_t h; void* base; };

}  // namespace

struct StubComm {
    int rank = 0, size = 1, id = 0;
    bool rendezvous = false;          // made by ncclCommInitRank with n > 1
    bool cross = false;               // some peer is another process
    Seg* seg = nullptr;
    hipEvent_t ready = nullptr, pulled = nullptr;
    std::vector<Opened> opened[MAX_RANKS];
};

namespace {

struct Pending { const void* send; void* recv; size_t bytes; StubComm* comm; hipStream_t stream; };
std::vector<Pending> g_pending;      // mode (1): calls made inside a group
int g_group_depth = 0;
std::atomic<int> g_next_id{1}, g_allgathers{0}, g_groups{0};

void shm_name(const ncclUniqueId& id, char out[64]) {
    uint64_t k[2];
    memcpy(k, id.internal, sizeof k);
    snprintf(out, 64, "/h2agg_rccl_standin_%016llx%016llx", (unsigned long long)k[0], (unsigned long long)k[1]);
}

// sense-counting barrier over the segment; false after TIMEOUT_S (a rank never came) or when a peer already gave up
bool barrier(StubComm* c) {
    Seg* s = c->seg;
    if (s->broken.load()) return false;
    const uint32_t gen = s->gen.load();
    if (s->count.fetch_add(1) + 1 == (uint32_t)c->size) {
        s->count.store(0);
        s->gen.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; s->gen.load() == gen; ++spin) {
        if (s->broken.load()) return false;
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
        if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(TIMEOUT_S)) {
            s->broken.store(1);
            return false;
        }
    }
    return true;
}

bool make_events(StubComm* c) {
    return hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c->pulled, hipEventDisableTiming) == hipSuccess;
}

// mode (1) and one-rank communicators: everything is known to this thread, no rendezvous.  recv of every rank =
// [rank 0's send | rank 1's send | ...], each rank's copies on its own stream behind the senders' events.
ncclResult_t flush() {
    std::vector<Pending> p;
    p.swap(g_pending);
    for (const Pending& a : p)
        if ((int)p.size() < a.comm->size) return 3;   // a rank did not call: a real RCCL would hang
    for (const Pending& a : p)
        if (hipEventRecord(a.comm->ready, a.stream) != hipSuccess) return 1;
    for (const Pending& dst : p) {
        for (const Pending& src : p) {
            if (src.comm->id != dst.comm->id) continue;
            if (src.bytes != dst.bytes) return 1;
            if (&src != &dst && hipStreamWaitEvent(dst.stream, src.comm->ready, 0) != hipSuccess) return 1;
            if (hipMemcpyAsync((char*)dst.recv + src.bytes * src.comm->rank, src.send, src.bytes, hipMemcpyDeviceToDevice, dst.stream) != hipSuccess)
                return 1;
        }
        if (hipEventRecord(dst.comm->pulled, dst.stream) != hipSuccess) return 1;
    }
    for (const Pending& a : p)   // a sender's stream may refill its send buffer only when every peer has read it
        for (const Pending& b : p)
            if (&a != &b && a.comm->id == b.comm->id && hipStreamWaitEvent(a.stream, b.comm->pulled, 0) != hipSuccess) return 1;
    return 0;
}

void* open_peer(StubComm* c, int q, const Slot& s) {
    for (const Opened& o : c->opened[q])
        if (memcmp(&o.h, &s.handle, sizeof s.handle) == 0) return o.base;
    void* base = nullptr;
    if (hipIpcOpenMemHandle(&base, s.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return nullptr;
    c->opened[q].push_back({s.handle, base});
    return base;
}

// modes (2) and (3)
ncclResult_t rendezvous_allgather(const void* send, void* recv, size_t bytes, StubComm* c, hipStream_t s) {
    Slot& mine = c->seg->slot[c->rank];
    if (hipEventRecord(c->ready, s) != hipSuccess) return 1;   // the send buffer is what the stream has made of it by HERE
    mine.send = (uint64_t)send;
    mine.bytes = bytes;
    mine.ev_ready = (uint64_t)c->ready;
    mine.ev_pulled = (uint64_t)c->pulled;
    if (c->cross) {
        if (hipStreamSynchronize(s) != hipSuccess) return 1;   // (events do not cross processes here)
        void* base = nullptr;
        size_t span = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &span, (hipDeviceptr_t)send) != hipSuccess) return 1;
        if (hipIpcGetMemHandle(&mine.handle, base) != hipSuccess) return 1;
        mine.off = (uint64_t)((const char*)send - (const char*)base);
    }
    if (!barrier(c)) return 3;                                 // every rank has made this call
    const int pid = (int)getpid();
    for (int q = 0; q < c->size; ++q) {
        const Slot& sq = c->seg->slot[q];
        if (sq.bytes != bytes) return 1;
        const void* src;
        if (sq.pid == pid) {
            src = (const void*)sq.send;
            if (q != c->rank && hipStreamWaitEvent(s, (hipEvent_t)sq.ev_ready, 0) != hipSuccess) return 1;
        } else {
            void* base = open_peer(c, q, sq);
            if (!base) return 1;
            src = (const char*)base + sq.off;
        }
        if (hipMemcpyAsync((char*)recv + bytes * q, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    }
    if (hipEventRecord(c->pulled, s) != hipSuccess) return 1;
    if (c->cross && hipStreamSynchronize(s) != hipSuccess) return 1;
    if (!barrier(c)) return 3;                                 // every rank has enqueued (cross-process: finished) its pulls
    for (int q = 0; q < c->size; ++q) {
        const Slot& sq = c->seg->slot[q];
        if (q != c->rank && sq.pid == pid && hipStreamWaitEvent(s, (hipEvent_t)sq.ev_pulled, 0) != hipSuccess) return 1;
    }
    return 0;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0x5a, sizeof *id);
    uint64_t k[2] = {(uint64_t)getpid() << 32 | (uint32_t)g_next_id.fetch_add(1),
                     (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count()};
    memcpy(id->internal, k, sizeof k);
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
    if (n < 1 || n > MAX_RANKS || rank < 0 || rank >= n) return 5;
    StubComm* c = new StubComm;
    c->rank = rank;
    c->size = n;
    c->id = g_next_id.fetch_add(1);
    if (!make_events(c)) return 1;
    if (n > 1) {   // like RCCL's, this call returns when every rank of the id has made it
        char name[64];
        shm_name(id, name);
        const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Seg)) != 0) return 1;   // (same size from every rank; new pages read as zero)
        c->seg = (Seg*)mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (c->seg == MAP_FAILED) return 1;
        c->rendezvous = true;
        c->seg->slot[rank].pid = (int)getpid();
        const bool ok = barrier(c);
        if (rank == 0) shm_unlink(name);                          // the mappings keep it alive; nothing is left in /dev/shm
        if (!ok) return 3;
        for (int q = 0; q < n; ++q) c->cross |= c->seg->slot[q].pid != (int)getpid();
        if (!barrier(c)) return 3;
    }
    *out = c;
    return 0;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
    const int id = g_next_id.fetch_add(1);
    for (int i = 0; i < n; ++i) {
        comms[i] = new StubComm;
        comms[i]->rank = i;
        comms[i]->size = n;
        comms[i]->id = id;
        if (!make_events(comms[i])) return 1;
    }
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    for (auto& v : c->opened)
        for (Opened& o : v) (void)hipIpcCloseMemHandle(o.base);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->pulled) (void)hipEventDestroy(c->pulled);
    if (c->seg) munmap(c->seg, sizeof(Seg));
    delete c;
    return 0;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->size; return 0; }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
    g_allgathers.fetch_add(1);
    if (c->rendezvous) return rendezvous_allgather(send, recv, count, c, s);
    g_pending.push_back({send, recv, count, c, s});   // (mode (1) is single-threaded by construction)
    return g_group_depth ? 0 : flush();
}

ncclResult_t ncclGroupStart() { ++g_group_depth; g_groups.fetch_add(1); return 0; }
ncclResult_t ncclGroupEnd() { return --g_group_depth == 0 ? flush() : 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : r == 3 ? "stand-in: a rank of the communicator did not call" : "stand-in: failure"; }
// for the drivers' assertions
int rccl_stub_allgathers() { return g_allgathers.load(); }
int rccl_stub_groups() { return g_groups.load(); }

}  // extern "C"
