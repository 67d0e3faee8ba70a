// libh2agg.so — context management and the C ABI declared in include/h2agg.h.
// Everything numeric runs in the HIP kernels of batch_kernels.hpp / msm_kernels.hpp; there is no CPU
// arithmetic path in this library (a context cannot be created without a HIP device).
#include "../../include/h2agg.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <csignal>
#include <dlfcn.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <sched.h>
#include <sys/resource.h>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "scalar_mul_kernels.hpp"
#include "lp_kernels.hpp"
#include "pairing.hpp"
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
// the same pairing once more, compiled for BMI2 + ADX (csrc/pairing.hpp's header): taken when the CPU has both
#define H2AGG_PAIRING_ADX_BUILD 1
#define PAIRING_NS pairing_adx
#define PAIRING_ADX 1
#pragma clang attribute push(__attribute__((target("bmi2,adx"))), apply_to = function)
#include "pairing.hpp"
#pragma clang attribute pop
#undef PAIRING_NS
#undef PAIRING_ADX
#endif
#include "poseidon_host.hpp"
#include "poseidon_sponge_host.hpp"
#include "poseidon_ifma_host.hpp"
#include "poseidon_kernels.hpp"

using namespace h2agg;

namespace {

// Experiment switches (occupancy caps, kernel variants, priorities, fault injection ...) exist only in a library built with
// -DH2AGG_MEASURE_KNOBS (tools/ and profiles/ say which runs used one).  A shipped library reads the environment for
// nothing but what include/h2agg.h documents under "Environment": H2AGG_HOST_THREADS, H2AGG_TRANSCRIPT, H2AGG_HOST_SPONGE,
// H2AGG_NO_PLACE, H2AGG_TRACE, H2AGG_TRACE_PHASES.  Per-call test hooks go through h2agg_debug_configure.
#ifdef H2AGG_MEASURE_KNOBS
static inline const char* knob(const char* name) { return getenv(name); }
#else
static inline const char* knob(const char*) { return nullptr; }
#endif
// Events that only order streams of this device / time kernels on it: device-scope release.  (The default is a system-scope
// release — an L2 write-back at every record.)  H2AGG_EVENT_SCOPE=system (a measure knob) restores the default for A/B runs.
static const bool ev_scope_system = knob("H2AGG_EVENT_SCOPE") && !strcmp(knob("H2AGG_EVENT_SCOPE"), "system");
#define EV_SYNC_FLAGS (ev_scope_system ? hipEventDisableTiming : (hipEventDisableTiming | hipEventReleaseToDevice))
#define EV_TIME_FLAGS (ev_scope_system ? hipEventDefault : hipEventReleaseToDevice)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

enum Stage {
    ST_PART_COUNT = 0, ST_PART_SCATTER, ST_BUCKET_SORT, ST_ORDER, ST_ACCUM, ST_ACCUM_BIG, ST_REDUCE, ST_WINDOW_SUM,
    ST_FINAL, ST_N
};
const char* const STAGE_NAMES[ST_N] = {"msm_part_count", "msm_part_scatter",   "msm_bucket_sort",
                                       "msm_order",      "msm_accumulate",     "msm_accumulate_big",
                                       "msm_reduce",     "msm_window_sum",     "msm_final"};

struct Table {
    uint8_t* d = nullptr;       // Montgomery affine, 64 B / point
    uint8_t* endo_x = nullptr;  // beta * x, 32 B / point (made on the first GLV MSM over this table)
    size_t n = 0;
    // fixed-base levels (h2agg_bases_precompute): pre[(w * n + i) * 64] = 2^(pre_c * w) * P_i, w < pre_W
    uint8_t* pre = nullptr;
    int pre_c = 0, pre_W = 0;
    // comb of the first comb_n bases (k_comb_table_build with bases; made on the first small MSM over a table that has
    // fixed-base levels, i.e. one its owner declared constant): comb[((b * 32 + w) * 255 + d - 1) * 64] = d * 2^(8w) * P_b
    uint8_t* comb = nullptr;
    size_t comb_n = 0;
    std::vector<uint8_t*> comb_retired;   // smaller combs this one replaced: kernels in flight may still read them, so they
                                          // are freed where the table's other buffers are (behind a synchronisation)
    void free_combs() {
        if (comb) hipFree(comb);
        for (uint8_t* p : comb_retired) hipFree(p);
        comb = nullptr;
        comb_n = 0;
        comb_retired.clear();
    }
};
struct PreTable {   // what msm_run needs of it
    const uint8_t* d;
    size_t n_level;
    int c, W;
};

}  // namespace

constexpr int MSM_MAX_SLICES = 16;
enum { CHAIN_OFF = 0, CHAIN_FIRST = 1, CHAIN_MID = 2, CHAIN_LAST = 3 };

struct h2agg_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::string desc;
    int cu_count = 0;

    // grow-only device workspace
    DevBuf in_a, in_b, in_c, out, tmp_bases;                      // host-buffer entry points
    DevBuf comb;                                                  // fixed-base comb table of the generator (k_comb_table_build)
    bool comb_ready = false;
    // transcript side (csrc/transcript.inc): Poseidon constants as device registers; staging of proofs / items; the
    // decompressed points, element streams and challenges of the last transcript batch
    DevBuf psd_spec, tr_in, tr_points, tr_elems, tr_chal;
    bool psd_ready = false;
    int cfg_transcript = 0;            // sponge backend: 0 = auto, 1 = device, 2 = host threads (h2agg_transcript_configure)
    std::vector<uint8_t> h_elems;      // element streams downloaded for the host backend
    // verifier pipeline (csrc/verifier.inc): instance values / commitments of a circuit's proofs, the aggregation transcript
    DevBuf inst_vals, inst_jac, inst_aff, agg_elems;
    DevBuf hist[2], offs[2], pmeta[2], order[2], entries[2];   // what the accumulation reads: one set per sort slot (overlap level 3)
    DevBuf big_list[2], big_keys[2], big_part[2];              // written by the accumulation, read by the over-long-bucket kernels
    DevBuf fix_list[2];                                        // buckets the lean accumulation left to the general formulas
    bool meta_clean[2] = {};                                   // pmeta[q] was zeroed behind its last use (tail stream)
    DevBuf item_idx, item_sub,
        glv_buf, parts, small, endo_buf, tile_counts, fb_long;  // MSM (bulk side: main stream only)
    uint32_t* d_flags = nullptr;      // in `small`: [0] status flags, [1] big_count
    uint8_t* d_res_xyzz = nullptr;    // in `small` + 1024 + 144 * slot of the LAST msm_run (see msm_run)
    uint8_t* d_res_jac = nullptr;     // in `small` + 256
    uint8_t* h_pinned = nullptr;      // 4 KiB pinned staging for small results / flags

    // schema-layer scratch (grow-only; the schema API is synchronous, one evaluation is in flight per context):
    // the Fr register file, the uploaded staging block, per-side MSM scalars / Montgomery bases, pinned staging
    DevBuf sch_regs, sch_in, sch_scalars[2], sch_bases[2], sch_endo;
    uint8_t* h_stage = nullptr;
    size_t h_stage_cap = 0;
    uint8_t* h_down = nullptr;        // page-locked landing block of the from-bytes call's downloads (csrc/verifier.inc)
    size_t h_down_cap = 0;
    const void* sch_owner = nullptr;  // the schema whose tape sch_regs reflects
    struct AggPlanCache* agg_plans = nullptr;   // recorded aggregations kept for the next call of the same shape (csrc/verifier.inc)

    // host-buffer MSM: slices are copied on this stream while the previous slice is computed
    hipStream_t copy_stream = nullptr;
    hipStream_t spare_streams[4] = {};   // what place_streams() did not give a role (see there)
    // verifier pipeline: point decompression of a circuit's proofs runs here, beside the instance-column MSMs
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_aux = nullptr, ev_aux_go = nullptr;
    hipEvent_t ev_copy[MSM_MAX_SLICES] = {}, ev_copy_s[MSM_MAX_SLICES] = {}, ev_ready = nullptr;

    std::map<uint64_t, Table> tables;
    uint64_t next_handle = 1;

    // multi-GPU: this context's rank in an RCCL communicator (csrc/comm.inc); ncclComm_t kept opaque here
    void* comm = nullptr;
    int comm_rank = 0, comm_size = 0;

    // h2agg_debug_configure: test hooks read per call (chained host-buffer slices, comb route, plan cache)
    int dbg_pcie_slices = 0, dbg_pcie_glv = 0, dbg_pcie_chain = 1, dbg_comb_msm = 1, dbg_plan_cache = 1, dbg_small_sort = 1, dbg_eval_split = 1, dbg_pre_big = 0, dbg_lean_acc = 1, dbg_shard_fail = 0, dbg_shard_calls = 0, dbg_phases = 0, dbg_tape_lds = 1, dbg_prewake = 1;
    std::string last_phases;   // debug key phases: the last h2agg_verify_aggregation's wall-clock split (h2agg_last_phases)
    // tuning
    int cfg_c = 0, cfg_seg = 0, cfg_big = 0, cfg_sub_bits = 0, cfg_tile = 0;
    int cfg_glv = 0;   // 0 = auto, 1 = on, -1 = off
    int cfg_lpb = 0;   // lanes per bucket in the accumulate kernel: 0 = auto, 1 / 2 / 4 / 8 / 16
    bool cfg_no_stage = false, cfg_stage_l1 = false, staged_attr_set = false, cfg_no_dm = false, dm_attr_set = false, r2d_attr_set = false;

    // optional overlap of the serial tail (k_msm_final) of MSM k with the bulk of MSM k+1
    bool tail_overlap = false;
    int overlap_level = 2;  // 1: only the Horner tail on the second stream; 2: reduction + window sums + tail
    // TAIL_SLOTS tail streams used round-robin, each with its own buckets / segsum / wsum set: the latency-shaped
    // tails of consecutive MSMs overlap one another as well as the next MSMs' bulk (small MSMs are otherwise
    // bound by one tail chain: ~0.7 ms per MSM at 2^14 points against 0.5 ms of bulk).
    static constexpr int TAIL_SLOTS = 3;
    // what a tail reads: one independent allocation per slot, so MSMs with DIFFERENT plans (bucket counts, segment
    // counts, window counts) in flight on different slots can never alias one another, whatever their sizes
    DevBuf r2d_ticket[TAIL_SLOTS];
    DevBuf buckets[TAIL_SLOTS], segsum[TAIL_SLOTS], wsum[TAIL_SLOTS];
    hipStream_t tail_streams[TAIL_SLOTS] = {};
    hipEvent_t ev_bulk[TAIL_SLOTS] = {}, ev_tail[TAIL_SLOTS] = {};
    bool tail_pending[TAIL_SLOTS] = {};
    int parity = 0;   // slot of the next MSM
    // Slices of ONE MSM that share a bucket set (an MSM is a sum over points: the slices' bucket sums just add up, so only
    // the last slice needs the bucket reduction / window sums / Horner tail — one latency chain per MSM instead of one per
    // slice).  0 = off; CHAIN_FIRST / CHAIN_MID: sort + accumulation only, the slot does not rotate; CHAIN_LAST: accumulation
    // on top of what the slot holds, then the tail.  chain_n: the point count the plan is made for (every slice the same
    // plan); chain_glv: the plan's endomorphism split.
    int chain = 0;
    size_t chain_n = 0;
    bool chain_glv = false;
    // called by msm_run on the stream of the accumulation, behind the sort and in front of the first kernel that reads the
    // bases: the host-buffer MSM waits there for the slice's bases (the sort only needs the scalars, which cross PCIe first)
    std::function<int(hipStream_t)> bases_hook;
    // overlap level 3: the bucket accumulation of MSM k runs on its own stream, under the sort of MSM k+1 (which stays on the
    // context's stream, ordered after whatever the caller queued there).  The sort's outputs exist twice.
    // Deferred tail (overlap level >= 2): the tail of MSM k is launched from the NEXT MSM's call, behind that MSM's sort —
    // beside the sort it slows the sort's latency chains by 2x, beside the accumulation it only costs its own instructions.
    // Whoever needs the results first (join_tails) launches it at once.
    std::function<int(bool)> deferred_tail;   // argument: true = wait for the event just recorded behind a sort
    hipEvent_t ev_sortdone = nullptr;
    hipStream_t acc_stream = nullptr;
    hipEvent_t ev_sorted[2] = {}, ev_accdone[2] = {};
    bool accdone_pending[2] = {};
    int sort_par = 0;

    // profiling: a ring of per-call event sets, harvested lazily so that measuring does not serialise
    // back-to-back asynchronous MSMs
    static constexpr int PROF_RING = 32;
    struct ProfSlot {
        hipEvent_t ev[ST_N][2] = {};
        bool used[ST_N] = {};
        bool pending = false;
    };
    bool profiling = false;
    int prof_only = -1;   // >= 0: only this stage is bracketed (one event pair per MSM instead of nine)
    bool prof_events_created = false;
    ProfSlot prof[PROF_RING];
    int prof_cur = 0;
    double stage_ms[ST_N] = {};
    uint64_t stage_launches[ST_N] = {};
};

namespace {

int fail(h2agg_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(ctx, H2AGG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

int flush_deferred_tail(h2agg_ctx* c, bool behind_sort);

int ensure(h2agg_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return H2AGG_OK;
    if (b.p) {
        // a grow-only buffer is replaced: nothing queued on ANY stream of the device may still refer to the old one
        // (nor anything not yet queued: a deferred tail goes out first)
        {
            const int rc_ = flush_deferred_tail(c, false);
            if (rc_ != H2AGG_OK) return rc_;
        }
        HIP_TRY(c, hipDeviceSynchronize());
        HIP_TRY(c, hipFree(b.p));
    }
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return fail(c, H2AGG_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    b.cap = want;
    return H2AGG_OK;
}

#define TRY(expr)                      \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != H2AGG_OK) return rc_; \
    } while (0)

int grid_for(const h2agg_ctx* c, size_t n) {
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    size_t cap = (size_t)c->cu_count * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// d_flags: [0] status of the call in progress, [1] spare, [2] sticky status of earlier asynchronous calls.
// A synchronous entry point starts with clear_flags (which ROLLS [0] into [2], so a flag raised by a preceding
// h2agg_g1_msm_device_async / _batch_async is never lost) and ends with finish (which reports both).
int clear_flags(h2agg_ctx* c) {
    hipLaunchKernelGGL(k_flags_roll, dim3(1), dim3(1), 0, c->stream, c->d_flags);
    return H2AGG_OK;
}

int flags_to_status(h2agg_ctx* c, uint32_t f, bool earlier) {
    const char* pre = earlier ? "an earlier asynchronous call on this context: " : "";
    if (f & FLAG_DIV_ZERO)
        return fail(c, H2AGG_ERR_DIV_ZERO, std::string(pre) + "inversion of zero (reference: invert().unwrap() panics)");
    if (f & FLAG_NONCANONICAL) return fail(c, H2AGG_ERR_NONCANONICAL, std::string(pre) + "input integer >= modulus");
    if (f & FLAG_BAD_POINT) return fail(c, H2AGG_ERR_BAD_POINT, std::string(pre) + "invalid point encoding in proof");
    return H2AGG_OK;
}

// synchronise and translate device status flags (this call's, then any left behind by earlier asynchronous calls)
// host_flags: FLAG_* bits raised by host-side halves of the call (the host sponge's canonicity check): merged into the
// call's device flags so that the order of precedence of the errors does not depend on where a check ran
int finish(h2agg_ctx* c, uint32_t host_flags = 0) {
    HIP_TRY(c, hipMemcpyAsync(c->h_pinned + 2048, c->d_flags, 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t f[3];
    memcpy(f, c->h_pinned + 2048, 12);
    f[0] |= host_flags;
    if (f[0] | f[2]) HIP_TRY(c, hipMemsetAsync(c->d_flags, 0, 12, c->stream));   // reported once, here
    if (f[0]) return flags_to_status(c, f[0], false);
    return flags_to_status(c, f[2], true);
}

// bits of the scalar that fall into the top window (a narrow top window means a few huge buckets).
// plain: 254-bit scalars, W*c >= 255; GLV: 127-bit magnitudes, W*c >= 128
int window_count(int c, bool glv) { return glv ? (128 + c - 1) / c : (255 + c - 1) / c; }
int top_window_bits(int c, bool glv) { return (glv ? 127 : 254) - c * (window_count(c, glv) - 1); }
int choose_window(size_t n, bool glv) {
    // Measured on MI355X (tools/window_sweep.py, profiles/r01_window_sweep_*.txt).  Two things decide it, and the
    // classic "log2(n) - 4" rule sees neither: (1) the tail (bucket reduce -> window sums -> Horner) is a serial
    // chain whose length grows with the number of windows, so below ~2^17 points FEWER, WIDER windows win even
    // though they leave most buckets empty; (2) only widths whose top window is as sparse as the others
    // (GLV 8/13/16, plain 8/15/16: top window as sparse as the others) avoid a dense top window that costs as much as all the
    // other windows together.
    // (re-measured in round 4 behind the one-launch sort and the limb-parallel Horner chain, profiles/r04_sweeps.txt section 5:
    // 8 bits up to 2^12 points — 2^11: 0.457 against 0.490 ms alone, 0.194 against 0.250 back to back; 2^12 equal; the two
    // multi_exps of an evaluation of 4 proofs, 1 750 pairs: 0.68 against 0.76 ms)
    if (glv) return n <= ((size_t)1 << 12) ? 8 : n <= ((size_t)1 << 14) ? 13 : 16;
    // 17 bits from 1.5 * 2^20 points on: a bucket costs 2 general additions (~2.8 insertions) whatever the point count, a
    // window's insertions grow with it — 15 windows of 2^16 buckets overtake 16 of 2^15 between 2^20 points (+0.7 %) and
    // 2^21 (+5.7 %; 2^22: +8.8 %, profiles/r03_sweeps.txt section 11)
    return n <= ((size_t)1 << 12) ? 8 : n < ((size_t)1 << 19) ? 15 : n < ((size_t)3 << 19) ? 16 : 17;
}

MsmPlan make_plan(const h2agg_ctx* c, size_t n, uint32_t batch = 1) {
    MsmPlan p;
    // GLV halves the latency-shaped stages (reduction, Horner tail) at the price of the decomposition pass and the
    // slice-combine pass of the half-as-many buckets: a win whenever those stages are exposed — single-MSM latency
    // mode, or small / medium MSMs — and a small loss when a large MSM's tail is hidden under the next one's bulk
    // (profiles/r01_sweeps.txt), or when the tail is a small share anyway.  auto = on unless (overlap mode and n >= 2^20) or
    // n >= 2^22.
    p.glv = c->cfg_glv > 0 ||
            (c->cfg_glv == 0 && !(c->tail_overlap && n >= ((size_t)1 << 20)) && n < ((size_t)1 << 22));
    p.c = c->cfg_c ? c->cfg_c : choose_window(n, p.glv);
    p.W = window_count(p.c, p.glv);
    p.NB = 1u << (p.c - 1);
    p.NBT = (uint32_t)p.W * batch * p.NB;   // a batch of MSMs over one table is one MSM with batch x W windows
    // segment length of the bucket reduction: ~1024 waves of running sums (profiles/r01_sweeps.txt)
    // (short segments = short chains; when the tail is hidden under the next MSM's bulk its WORK is what costs, and the
    // double-and-add by the segment offset — as much work as 8 buckets of running sums — amortises over longer segments)
    uint32_t seg = c->cfg_seg ? (uint32_t)c->cfg_seg
                              : (p.NBT >= (1u << 18) ? (c->tail_overlap && p.NBT >= (1u << 19) ? 32u : 8u)
                                                     : (p.NBT >= (1u << 15) ? 4u : 2u));
    if (seg > p.NB) seg = p.NB;
    p.seg = seg;
    p.spw = p.NB / seg;
    p.big = c->cfg_big ? (uint32_t)c->cfg_big : 256u;
    return p;
}

struct StageTimer {
    h2agg_ctx* c;
    int st;
    hipStream_t s;
    StageTimer(h2agg_ctx* c_, int st_, hipStream_t s_ = nullptr) : c(c_), st(st_), s(s_ ? s_ : c_->stream) {
        if (c->profiling && (c->prof_only < 0 || c->prof_only == st)) {
            hipEventRecord(c->prof[c->prof_cur].ev[st][0], s);
        }
    }
    ~StageTimer() {
        if (c->profiling && (c->prof_only < 0 || c->prof_only == st)) {
            hipEventRecord(c->prof[c->prof_cur].ev[st][1], s);
            c->prof[c->prof_cur].used[st] = true;
        }
    }
};

// Debugging aid (H2AGG_BREADCRUMBS=1): the return addresses of the last C-ABI entries (every entry point binds its context
// first) and the sizes of the last MSMs, dumped when the process aborts — the HSA runtime abort()s on a GPU memory fault, long
// after the host call that queued the faulting kernel has returned.  Resolve the addresses against libh2agg.so (they are
// printed relative to its load address).
struct Crumbs {
    static constexpr int N = 48;
    std::atomic<uint32_t> at{0};
    uintptr_t addr[N] = {};
    uint64_t note[N] = {};
    bool on = false;
};
Crumbs g_crumbs;
void crumbs_dump(int) {
    Dl_info info;
    uintptr_t base = 0;
    if (dladdr((void*)&crumbs_dump, &info)) base = (uintptr_t)info.dli_fbase;
    const uint32_t end = g_crumbs.at.load();
    fprintf(stderr, "[h2agg breadcrumbs pid %d] last entries, oldest first (addresses relative to libh2agg.so):\n", (int)getpid());
    for (uint32_t k = end > Crumbs::N ? end - Crumbs::N : 0; k < end; ++k)
        fprintf(stderr, "  #%u  +0x%zx  note %llu\n", k, (size_t)(g_crumbs.addr[k % Crumbs::N] - base), (unsigned long long)g_crumbs.note[k % Crumbs::N]);
    fflush(stderr);
    signal(SIGABRT, SIG_DFL);
    abort();
}
inline void crumb(uintptr_t a, uint64_t note) {
    if (!g_crumbs.on) return;
    const uint32_t k = g_crumbs.at.fetch_add(1) % Crumbs::N;
    g_crumbs.addr[k] = a;
    g_crumbs.note[k] = note;
}
struct CrumbsInit {
    CrumbsInit() {
        if (knob("H2AGG_BREADCRUMBS")) {
            g_crumbs.on = true;
            signal(SIGABRT, crumbs_dump);
        }
    }
} g_crumbs_init;

// Debugging aid (H2AGG_CHAOS=<bits>, off by default): a one-lane kernel that just waits, put in front of work on a stream to
// shift the streams against one another — a missing event dependency then shows as a wrong result in the test suites instead of
// as a once-in-twenty-runs memory fault.  bit 0: in front of every tail; bit 1: in front of every accumulation; bit 2: in
// front of every sort; bit 3: in front of the auxiliary stream's transcript work (verifier.inc); bit 4: in front of every
// slice's copies of the host-buffer MSM.
__global__ void k_chaos_wait(uint32_t us) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)us * 100u) __builtin_amdgcn_s_sleep(32);   // wall_clock64 ticks at 100 MHz
}
inline int chaos_bits() {
    static const int bits = knob("H2AGG_CHAOS") ? atoi(knob("H2AGG_CHAOS")) : 0;
    return bits;
}
inline void debug_sync(int bit) {   // H2AGG_SYNC_AT=<bits>: a device-wide wait at that point of msm_run (bisecting a race)
    static const int bits = knob("H2AGG_SYNC_AT") ? atoi(knob("H2AGG_SYNC_AT")) : 0;
    if (bits & bit) hipDeviceSynchronize();
}
inline void chaos_wait(int bit, hipStream_t s, uint32_t us = 300) {
    if (chaos_bits() & bit) hipLaunchKernelGGL(k_chaos_wait, dim3(1), dim3(1), 0, s, us);
}

// Which of the context's streams may run BESIDE the main stream?  HIP gives a stream its hardware queue when it is first used,
// round robin over GPU_MAX_HW_QUEUES (default 4) queues, so which streams share a queue depends on everything the process used
// before — and a tail stream on the main stream's queue puts its latency chains (0.5 ms kernels of one wave) IN FRONT of the next
// MSM's sort and accumulation: back-to-back 2^20-point MSMs then take 1.55-1.65 instead of 1.25 ms (profiles/r03_sweeps.txt
// section 18; a lazily made copy stream on that queue cost the host-buffer MSM 0.9 ms).  So the roles are handed out by
// measurement: every candidate is kept busy with a waiting kernel in turn while a tiny kernel goes to the main stream; the
// candidates that hold it up share its queue and get no role that matters.  ~4 ms per call; h2agg_create and h2agg_set_stream.
int place_streams(h2agg_ctx* c) {
    if (getenv("H2AGG_NO_PLACE")) return H2AGG_OK;
    constexpr int NP = h2agg_ctx::TAIL_SLOTS + 3 + 4;
    hipStream_t pool[NP];
    int np = 0;
    for (int k = 0; k < h2agg_ctx::TAIL_SLOTS; ++k) pool[np++] = c->tail_streams[k];
    pool[np++] = c->aux_stream;
    pool[np++] = c->copy_stream;
    for (int k = 0; k < 4; ++k) pool[np++] = c->spare_streams[k];
    pool[np++] = c->acc_stream;
    // first use (= queue assignment) in a fixed order: the main stream, then the candidates
    hipLaunchKernelGGL(k_chaos_wait, dim3(1), dim3(1), 0, c->stream, 1u);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < np; ++i) {
        hipLaunchKernelGGL(k_chaos_wait, dim3(1), dim3(1), 0, pool[i], 1u);
        HIP_TRY(c, hipStreamSynchronize(pool[i]));
    }
    bool blocks[NP];
    int nfree = 0;
    for (int i = 0; i < np; ++i) {
        int held = 0;   // (majority of three: one slow launch on a busy box must not cost a stream its role)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_chaos_wait, dim3(1), dim3(1), 0, pool[i], 250u);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_chaos_wait, dim3(1), dim3(1), 0, c->stream, 1u);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            HIP_TRY(c, hipStreamSynchronize(pool[i]));
            held += us > 120.0;
            if (rep == 1 && (held == 0 || held == 2)) break;
        }
        blocks[i] = held >= 2;
        nfree += !blocks[i];
    }
    if (knob("H2AGG_TRACE_STREAMS")) {
        fprintf(stderr, "[h2agg] streams that hold the main stream up when busy:");
        for (int i = 0; i < np; ++i) fprintf(stderr, " %d%s", i, blocks[i] ? "*" : "");
        fprintf(stderr, "  (* = shares its queue)\n");
    }
    if (nfree < h2agg_ctx::TAIL_SLOTS + 2) return H2AGG_OK;   // (not enough to choose from: keep the creation order)
    hipStream_t ordered[NP];
    int no = 0;
    for (int i = 0; i < np; ++i)
        if (!blocks[i]) ordered[no++] = pool[i];
    for (int i = 0; i < np; ++i)
        if (blocks[i]) ordered[no++] = pool[i];
    int at = 0;
    for (int k = 0; k < h2agg_ctx::TAIL_SLOTS; ++k) c->tail_streams[k] = ordered[at++];
    c->aux_stream = ordered[at++];
    c->copy_stream = ordered[at++];
    c->acc_stream = ordered[at++];
    for (int k = 0; k < 4; ++k) c->spare_streams[k] = ordered[at++];
    return H2AGG_OK;
}

// harvest one ring slot (blocks until that call's events have completed)
void profile_harvest(h2agg_ctx* c, int slot) {
    h2agg_ctx::ProfSlot& ps = c->prof[slot];
    if (!ps.pending) return;
    for (int s = 0; s < ST_N; ++s) {
        if (!ps.used[s]) continue;
        float ms = 0.f;
        hipEventSynchronize(ps.ev[s][1]);
        if (hipEventElapsedTime(&ms, ps.ev[s][0], ps.ev[s][1]) == hipSuccess) {
            c->stage_ms[s] += ms;
            c->stage_launches[s] += 1;
        }
        ps.used[s] = false;
    }
    ps.pending = false;
}
void profile_begin_call(h2agg_ctx* c) {
    if (!c->profiling) return;
    profile_harvest(c, c->prof_cur);  // slot about to be reused
}
void profile_end_call(h2agg_ctx* c) {
    if (!c->profiling) return;
    c->prof[c->prof_cur].pending = true;
    c->prof_cur = (c->prof_cur + 1) % h2agg_ctx::PROF_RING;
}
void profile_harvest_all(h2agg_ctx* c) {
    for (int k = 0; k < h2agg_ctx::PROF_RING; ++k) profile_harvest(c, k);
}

// make everything queued on the tail stream visible to the main stream
int flush_deferred_tail(h2agg_ctx* c, bool behind_sort) {
    if (!c->deferred_tail) return H2AGG_OK;
    std::function<int(bool)> f;
    f.swap(c->deferred_tail);
    return f(behind_sort);
}

int join_tails(h2agg_ctx* c) {
    TRY(flush_deferred_tail(c, false));
    for (int k = 0; k < h2agg_ctx::TAIL_SLOTS; ++k) {
        if (c->tail_pending[k]) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_tail[k], 0));
            c->tail_pending[k] = false;
        }
    }
    for (int q = 0; q < 2; ++q) {   // (every accumulation is followed by a tail, so these have completed: bookkeeping)
        if (c->accdone_pending[q]) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_accdone[q], 0));
            c->accdone_pending[q] = false;
        }
    }
    return H2AGG_OK;
}

// The MSM proper.  d_bases: Montgomery affine table; d_scalars: canonical 32-B scalars (device).
// Result: c->d_res_xyzz (Montgomery XYZZ) and, if d_out_jac != nullptr, canonical Jacobian there.
// batch > 1: `batch` MSMs over the SAME n bases, scalars laid out [batch][n]; results: canonical Jacobian at
// d_out_jac[96 * q] (c->d_res_xyzz then only holds MSM 0's XYZZ).  One set of launches does all of them: every
// scalar's windows are numbered q * W + w, and the stages after the sort only see batch * W windows.
// d_endo_x: beta * x per base (Table::endo_x) or nullptr = compute it here when the plan uses GLV.
// pre: fixed-base levels of the table (then d_bases is ignored): plain c-bit digits of ALL positions into one bucket set
// per MSM, bases looked up at level w; no Horner chain.
// can two MSMs of n_total points in all run as one split MSM (below)?  The one-launch sort's range, nothing forced.
bool msm_split_ok(const h2agg_ctx* c, size_t n_total) {
    if (!c->dbg_small_sort || n_total < 2 || n_total > (size_t)SMALL_SORT_N) return false;
    if (c->cfg_no_stage || c->cfg_stage_l1 || c->cfg_sub_bits || c->cfg_tile || c->chain) return false;
    const MsmPlan p = make_plan(c, n_total, 2);
    return p.NB <= (uint32_t)SMALL_SORT_NB;
}

// split != 0: TWO MSMs over the disjoint parts [0, split) and [split, n_base) of one table (scalars laid out alike), as one
// set of launches with 2 W windows; results: XYZZ at c->d_res_xyzz and c->d_res_xyzz + XYZZ_BYTES.  Only for sizes the
// one-launch sort takes (msm_split_ok): the evaluation's two multi_exps.  When the plan (make_plan(c, n_base, 2)) uses GLV,
// d_scalars are glv_decompose() words and d_endo_x is given (k_eval_prep wrote both); the tail stays on the context's
// stream — nothing follows that it could hide under, and a stream hand-over costs 10-20 us each way.
// the (level, point) sort of fb_sort_kernels.hpp has no sort / segment knobs: a context configured with any
// (h2agg_msm_configure_sort, reduce_segment) does not take it
bool fb_sort_knobs_clear(const h2agg_ctx* c) {
    return !c->cfg_no_dm && !c->cfg_no_stage && !c->cfg_stage_l1 && !c->cfg_sub_bits && !c->cfg_tile && !c->cfg_seg;
}

int msm_run(h2agg_ctx* c, const uint8_t* d_bases, const uint8_t* d_scalars, size_t n_base, uint8_t* d_out_jac,
            uint32_t batch = 1, const uint8_t* d_endo_x = nullptr, const PreTable* pre = nullptr, uint32_t split = 0) {
    if (split) batch = 2;
    const size_t n = split ? n_base : n_base * batch;   // scalars
    crumb((uintptr_t)__builtin_return_address(0), ((uint64_t)batch << 40) | n_base);
    if (n >= ((size_t)1 << 30)) return fail(c, H2AGG_ERR_INVALID, "n must be < 2^30");
    const int chain = (pre || batch != 1) ? CHAIN_OFF : c->chain;
    MsmPlan p = make_plan(c, chain ? c->chain_n : n_base, batch);
    if (chain && p.glv != c->chain_glv) {   // (the caller fixed the split for the whole chain)
        const int was = c->cfg_glv;
        c->cfg_glv = c->chain_glv ? 1 : -1;
        p = make_plan(c, c->chain_n, batch);
        c->cfg_glv = was;
    }
    int Wd = p.W;                             // digit positions per scalar (what the recoding loops over)
    // fixed-base levels at c = 20 (tables of 2^18 .. 2^22 points): the (level, point) sort of fb_sort_kernels.hpp
    const bool fbdm = pre && batch == 1 && pre->c == FB_C && pre->W == FB_W && pre->n_level <= ((size_t)FB_MAX_TILES * FB_T) &&
                      fb_sort_knobs_clear(c);
    const uint32_t fb_ntile = (uint32_t)((n + FB_T - 1) / FB_T);
    if (pre) {
        p.glv = false;
        p.c = pre->c;
        Wd = pre->W;
        p.W = 1;                              // one bucket set per MSM
        p.NB = 1u << (p.c - 1);
        p.NBT = fbdm ? FB_NBT : batch * p.NB;   // (the top digit's own slots: fb_sort_kernels.hpp)
        // measured (tools/instance_seg_sweep.py, 2^17-point columns): 16-bucket segments up to 8 MSMs per batch, 32 beyond
        p.seg = c->cfg_seg ? (uint32_t)c->cfg_seg : (batch >= 16 ? 32u : 16u);
        if (p.seg > p.NB) p.seg = p.NB;
        p.spw = p.NB / p.seg;
        d_bases = pre->d;
    }
    const int W1 = p.W;                       // windows (bucket sets) per MSM
    const uint32_t WT = (uint32_t)W1 * batch;  // windows in total
    const size_t nent = n * (size_t)Wd * (p.glv ? 2 : 1);   // bucket insertions
    if (!c->cfg_big) {
        // a lane walks a bucket alone up to `big` entries: 8x the mean keeps a denser top window (up to 4x the mean
        // when it holds c-2 bits) out of the workgroup-per-chunk path, whose LDS tree only pays for real outliers
        const size_t mean = nent / p.NBT;
        if (8 * mean > p.big) p.big = (uint32_t)(8 * mean);
    }
    if (nent >= ((size_t)1 << 32)) return fail(c, H2AGG_ERR_INVALID, "n * windows must be < 2^32");
    SortPlan sp;
    int want_sub = c->cfg_sub_bits ? c->cfg_sub_bits : SORT_SUB_BITS;
    if (pre && !c->cfg_sub_bits) {
        // one bucket set per MSM: keep >= 256 level-2 partitions (one workgroup each) so the sort still fills the chip
        while (want_sub > 4 && (((size_t)p.NB * batch) >> want_sub) < 256) --want_sub;
    }
    sp.sub_bits = (p.c - 1 < want_sub) ? p.c - 1 : want_sub;
    while ((WT * (p.NB >> sp.sub_bits)) > (uint32_t)SORT_MAX_PW && sp.sub_bits < p.c - 1 &&
           sp.sub_bits < SORT_MAX_SUB_BITS)
        ++sp.sub_bits;  // keep the level-1 partition count within its LDS counters
    sp.SB = 1u << sp.sub_bits;
    sp.ppw = p.NB >> sp.sub_bits;
    sp.PW = WT * sp.ppw;
    if (batch > 1 && !split) {
        sp.n_base = (uint32_t)n_base;
        sp.W1 = (uint32_t)W1;
    }
    if (pre) sp.pre_n = (uint32_t)pre->n_level;
    sp.tile = c->cfg_tile ? (uint32_t)c->cfg_tile : 2048u;
    // packed-item staged path: index field of 31 - sub_bits bits, a tile's keys must fit the LDS stage
    sp.glv = p.glv;
    const int idx_bits = (p.glv ? 30 : 31) - sp.sub_bits;   // packed item: sub | neg | (endo) | idx
    // items carry the BASE index (fixed-base mode: up to W * n_level of them)
    bool staged = !c->cfg_no_stage && (pre ? (size_t)pre->W * pre->n_level : n_base) <= ((size_t)1 << idx_bits);
    const size_t keys_per_scalar = (size_t)Wd * (p.glv ? 2 : 1);
    if (staged && c->cfg_stage_l1 && (size_t)sp.tile * keys_per_scalar > (size_t)STAGE_ITEMS_L1)
        sp.tile = (uint32_t)(STAGE_ITEMS_L1 / keys_per_scalar);
    if (staged && sp.tile < (uint32_t)BLOCK) staged = false;
    if (sp.PW > (uint32_t)SORT_MAX_PW) return fail(c, H2AGG_ERR_INVALID, "too many sort partitions");
    const uint32_t nseg_total = WT * p.spw;
    // 4 lanes per chain in the bucket reduction / window sums (latency) or 1 (least work): see msm_kernels.hpp
    static const int par4_env = knob("H2AGG_PAR4") ? atoi(knob("H2AGG_PAR4")) : 0;
    // measured: wins up to 16384 segments (c <= 13; also a 2^20-point MSM with 32-bucket segments in throughput mode,
    // where 1 024 waves of 94-addition chains would otherwise outlast the step), loses to the extra work above
    // (a split MSM is two such MSMs side by side, each within the range)
    const bool par4 = par4_env ? par4_env > 0 : nseg_total <= (split ? 32768u : 16384u);
    // pmeta words: pcount [PW] | pstart [PW + 1] | pcursor [PW] | bin_count [SIZE_BINS] | bin_start [SIZE_BINS + 1] |
    //              bin_cursor [SIZE_BINS] | big-bucket counters [2], each padded by 64 words
    constexpr uint32_t META_PW = DM_MAX_PW;   // (the digit-major path has up to 16 x 512 partitions)
    constexpr uint32_t M_PSTART = META_PW + 64, M_PCURSOR = M_PSTART + META_PW + 64,
                       M_BCOUNT = M_PCURSOR + META_PW + 64, M_BSTART = M_BCOUNT + SIZE_BINS + 64,
                       M_BCURSOR = M_BSTART + SIZE_BINS + 64, M_BIG = M_BCURSOR + SIZE_BINS + 64, M_WORDS = M_BIG + 64;
    // overlap level 3 (see h2agg_ctx::acc_stream): this MSM's accumulation leaves the context's stream
    const bool piped = c->tail_overlap && c->overlap_level >= 3;
    // overlap level >= 2: consecutive MSMs alternate between the two sets of sort outputs, so that what still reads one set
    // behind the accumulation (over-long-bucket kernels, zeroing of the counters: tail stream) never holds up the next sort
    const bool altbuf = c->tail_overlap && c->overlap_level >= 2;
    const int sq = altbuf ? c->sort_par : 0;
    {
        const size_t cap0 = c->pmeta[sq].cap;
        TRY(ensure(c, c->pmeta[sq], M_WORDS * 4));
        if (c->pmeta[sq].cap != cap0) c->meta_clean[sq] = false;
    }
    TRY(ensure(c, c->hist[sq], (size_t)p.NBT * 4));
    TRY(ensure(c, c->offs[sq], (size_t)p.NBT * 4));
    TRY(ensure(c, c->order[sq], (size_t)p.NBT * 4));
    // digit-major sort (sort_kernels.hpp): plain 16-bit windows over one table, 2^16 .. 2^22 points
    static const bool dm_env_off = knob("H2AGG_SORT") && !strcmp(knob("H2AGG_SORT"), "packed");
    const size_t dm_row = p.glv ? 2 * n : n;   // keys per window (GLV: both halves of a scalar land in the same 8 windows)
    // (c = 17, plain scalars: 15 windows of 2^16 buckets, 16-bit magnitude codes + sign / zero bit rows — k_dm_digits17, k_dm_partition<true>)
    const bool dm17 = p.c == 17 && !p.glv;
    const bool dm = !dm_env_off && !c->cfg_no_dm && !c->cfg_no_stage && !c->cfg_stage_l1 && !c->cfg_sub_bits && !c->cfg_tile && !pre && batch == 1 &&
                    (p.c == 16 || dm17) && dm_row >= ((size_t)1 << 16) && dm_row <= ((size_t)1 << 22);
    const uint32_t dm_nwin = p.glv ? 8u : (dm17 ? 15u : 16u);
    // the one-launch sort of small MSMs (sort_kernels.hpp k_small_sort); a forced sort configuration keeps its own kernels
    static const bool small_env_off = knob("H2AGG_SMALL_SORT") && !strcmp(knob("H2AGG_SMALL_SORT"), "0");
    const bool small_sort = !small_env_off && c->dbg_small_sort && !pre && !dm && n <= (size_t)SMALL_SORT_N && p.NB <= (uint32_t)SMALL_SORT_NB &&
                            !c->cfg_no_stage && !c->cfg_stage_l1 && !c->cfg_sub_bits && !c->cfg_tile;
    if (split && (!small_sort || split >= n_base)) return fail(c, H2AGG_ERR_INVALID, "split MSM outside the one-launch sort's range");
    DmPlan dp{};
    if (dm) {
        const size_t n = dm_row;   // (shadows the point count inside this block)
        dp.n_pts = p.glv ? (uint32_t)(dm_row / 2) : 0xffffffffu;
        dp.n = (uint32_t)n;
        dp.n_pad = dm17 ? (uint32_t)((n + 63) & ~(size_t)63) : (uint32_t)((n + 7) & ~(size_t)7);   // (17-bit windows: bit rows, 64 keys per word)
        dp.ntile = (uint32_t)((n + DM_T1 - 1) / DM_T1);
        dp.n_row = dp.ntile * (uint32_t)DM_T1;
        dp.ppw = dm17 ? 128 : 64;   // (level 2 keeps <= 512 buckets per partition in LDS: sub_bits <= 9)
        while (n / dp.ppw > 4096 && dp.ppw < (uint32_t)DM_MAX_PPW) dp.ppw *= 2;   // ~4 K keys per level-2 partition (8 K at 2^22)
        dp.sub_bits = p.c - 1;
        for (uint32_t q = dp.ppw; q > 1; q >>= 1) --dp.sub_bits;
        dp.SB = 1u << dp.sub_bits;
        dp.idx_bits = 31 - dp.sub_bits;
    }
    TRY(ensure(c, c->item_idx, fbdm ? (size_t)fb_ntile * FB_KEYS1 * 4 : dm ? (size_t)dm_nwin * dp.n_row * 4 : nent * 4));
    if (fbdm) TRY(ensure(c, c->fb_long, FB_LONG_WORDS * 4));   // very long partitions across workgroups (fb_sort_kernels.hpp)
    TRY(ensure(c, c->item_sub, fbdm ? 0 : dm ? (size_t)dm_nwin * dp.n_pad * 2 + (dm17 ? (size_t)2 * dm_nwin * (dp.n_pad / 8) : 0) : nent * 2));
    TRY(ensure(c, c->entries[sq], nent * 4));
    // buckets / segsum / wsum exist once per tail slot: in overlap mode the reduction of MSM k (tail stream)
    // runs while MSM k+1 fills the next slot's set.  (They were one allocation cut at par * this-plan's-size: two MSMs
    // with different plans in flight then overlapped — ADVICE r1.)
    const int par = c->parity;
    TRY(ensure(c, c->buckets[par], (size_t)p.NBT * XYZZ_BYTES));
    {   // (the two-dimensional reduction keeps 4096 partial sums per window there)
        const size_t r2d_records = p.NB == (uint32_t)(R2D_ROWS * R2D<7>::COLS) ? (size_t)WT * (R2D<7>::THREADS + 2)
                                   : p.NB == (uint32_t)(R2D_ROWS * R2D<8>::COLS) ? (size_t)WT * (R2D<8>::THREADS + 2) : 0;
        const size_t fb_records = fbdm ? (size_t)FB_R2D_WINDOWS * (R2D<7>::THREADS + 3) : 0;   // parts, halves, bucket sums
        TRY(ensure(c, c->segsum[par], std::max(std::max((size_t)nseg_total, r2d_records), fb_records) * XYZZ_BYTES));
    }
    TRY(ensure(c, c->wsum[par], (size_t)(fbdm ? FB_R2D_WINDOWS + 2 : WT) * XYZZ_BYTES));
    const size_t max_slots = nent / BIG_CHUNK + nent / ((size_t)p.big + 1) + 2;   // chunks of over-long buckets
    const size_t max_keys = nent / ((size_t)p.big + 1) + 2;
    TRY(ensure(c, c->big_list[sq], max_slots * 12));
    TRY(ensure(c, c->big_keys[sq], max_keys * 12));
    TRY(ensure(c, c->big_part[sq], max_slots * XYZZ_BYTES));
    uint32_t* meta = (uint32_t*)c->pmeta[sq].p;
    uint32_t *pcount = meta, *pstart = meta + M_PSTART, *pcursor = meta + M_PCURSOR;
    uint32_t *bin_count = meta + M_BCOUNT, *bin_start = meta + M_BSTART, *bin_cursor = meta + M_BCURSOR;
    uint32_t* hist = (uint32_t*)c->hist[sq].p;
    uint32_t* offs = (uint32_t*)c->offs[sq].p;
    uint32_t* order = (uint32_t*)c->order[sq].p;
    uint32_t* item_idx = (uint32_t*)c->item_idx.p;
    uint16_t* item_sub = (uint16_t*)c->item_sub.p;
    uint32_t* fb_long = (uint32_t*)c->fb_long.p;
    uint32_t* entries = (uint32_t*)c->entries[sq].p;
    c->d_res_xyzz = (uint8_t*)c->small.p + 1024 + 144 * par;   // each tail slot has its own XYZZ result
    if (split) c->d_res_xyzz = (uint8_t*)c->small.p + 2048 + 2 * 144 * par;   // ... or its own two
    uint8_t* buckets = (uint8_t*)c->buckets[par].p;
    uint8_t* segsum = (uint8_t*)c->segsum[par].p;
    uint8_t* wsum = (uint8_t*)c->wsum[par].p;
    uint32_t* big_list = (uint32_t*)c->big_list[sq].p;
    uint32_t* big_keys = (uint32_t*)c->big_keys[sq].p;
    uint8_t* big_part = (uint8_t*)c->big_part[sq].p;
    uint32_t* big_count = meta + M_BIG;   // [0] chunk slots, [1] multi-chunk buckets (zeroed with the rest of meta)
    hipStream_t st = c->stream;
    const unsigned ntiles = (unsigned)((n + sp.tile - 1) / sp.tile);
    // per-tile partition counts from the counting pass, read back by the packed scatter pass (same tiles)
    uint32_t* tile_counts = nullptr;
    if (fbdm) {
        TRY(ensure(c, c->tile_counts, (size_t)fb_ntile * (FB_NPART + 1) * 4));
        tile_counts = (uint32_t*)c->tile_counts.p;
    } else if (dm) {
        TRY(ensure(c, c->tile_counts, (size_t)dm_nwin * dp.ntile * (dp.ppw + 1) * 4));
        tile_counts = (uint32_t*)c->tile_counts.p;
    } else if (staged && !c->cfg_stage_l1) {
        TRY(ensure(c, c->tile_counts, (size_t)ntiles * sp.PW * 4));
        tile_counts = (uint32_t*)c->tile_counts.p;
    }
    const bool meta_was_clean = altbuf && c->meta_clean[sq];
    c->meta_clean[sq] = false;
    profile_begin_call(c);
    // the accumulation that last read this slot's sort outputs (two MSMs ago, or any earlier one for an MSM that does not
    // leave the stream: it uses slot 0 and scratch the accumulation stream may still read — beta*x column, slice sums)
    for (int q = 0; q < 2; ++q) {
        if (c->accdone_pending[q] && (!altbuf || q == sq || (p.glv && !d_endo_x))) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_accdone[q], 0));
            c->accdone_pending[q] = false;
        }
    }

    bool endo_late = false;   // (with a bases hook the beta*x column is computed behind it, in front of the accumulation)
    if (p.glv && !d_endo_x) {
        TRY(ensure(c, c->endo_buf, n_base * 32));
        if (c->bases_hook) endo_late = true;
        else
            hipLaunchKernelGGL(k_bases_endo_x, dim3(grid_for(c, n_base)), dim3(BLOCK), 0, st, d_bases, n_base,
                               (uint8_t*)c->endo_buf.p);
        d_endo_x = (const uint8_t*)c->endo_buf.p;
    }
    debug_sync(1);
    chaos_wait(4, st);
    if (p.glv && !split) {  // k = k1 + lambda*k2: the sort below reads the decomposed words instead of the scalars
        // (a split MSM's caller hands over the decomposed words and the beta * x column: k_eval_prep)
        TRY(ensure(c, c->glv_buf, n * 32));
        StageTimer t(c, ST_PART_COUNT);
        hipLaunchKernelGGL(k_glv_decompose, dim3(grid_for(c, n)), dim3(BLOCK), 0, st, d_scalars, n,
                           (uint8_t*)c->glv_buf.p, c->d_flags);
        d_scalars = (const uint8_t*)c->glv_buf.p;
    }
    if (small_sort) {
        StageTimer t(c, ST_BUCKET_SORT);
        // (the length-ordering pass below counts into the scratch words: then they are cleared as a whole)
        if (nent >= ((size_t)1 << 16) && !meta_was_clean) HIP_TRY(c, hipMemsetAsync(meta, 0, M_WORDS * 4, st));
        hipLaunchKernelGGL(k_small_sort, dim3(WT), dim3(SMALL_SORT_TB), 0, st, d_scalars, (uint32_t)n, p.c, Wd,
                           (batch > 1 && !split) ? (uint32_t)n_base : 0u, split, p.glv, p.NB, hist, offs, entries, c->d_flags,
                           big_count);
    } else if (fbdm) {
        {
            StageTimer t(c, ST_PART_SCATTER);
            if (!meta_was_clean) HIP_TRY(c, hipMemsetAsync(meta, 0, M_WORDS * 4, st));
            hipLaunchKernelGGL(k_fb_partition, dim3(fb_ntile), dim3(FB_TB1), 0, st, d_scalars, (uint32_t)n, pcount, tile_counts, item_idx,
                               c->d_flags);
            hipLaunchKernelGGL(k_fb_scan_list, dim3(1), dim3(1024), 0, st, (const uint32_t*)pcount, pstart, fb_ntile, fb_long);
        }
        {
            StageTimer t(c, ST_BUCKET_SORT);
            hipLaunchKernelGGL(k_fb_bucket_sort, dim3(FB_NPART), dim3(FB_TB2), 0, st, (const uint32_t*)pstart, (const uint32_t*)tile_counts,
                               (const uint32_t*)item_idx, (uint32_t)pre->n_level, fb_ntile, hist, offs, entries);
            // partitions too long for the stage (skewed / small scalars): the very long ones split over workgroups, the rest one
            // workgroup each; with uniform scalars every workgroup of the three launches leaves at once
            hipLaunchKernelGGL(k_fb_long_count, dim3(FB_LONG_CAP, FB_LONG_S), dim3(FB_TB2), 0, st, (const uint32_t*)pstart,
                               (const uint32_t*)tile_counts, (const uint32_t*)item_idx, (uint32_t)pre->n_level, fb_ntile, fb_long, hist, offs);
            hipLaunchKernelGGL(k_fb_long_place, dim3(FB_LONG_CAP, FB_LONG_S), dim3(FB_TB2), 0, st, (const uint32_t*)pstart,
                               (const uint32_t*)tile_counts, (const uint32_t*)item_idx, (uint32_t)pre->n_level, fb_ntile,
                               (const uint32_t*)fb_long, entries);
            hipLaunchKernelGGL(k_fb_bucket_sort_long, dim3(FB_NPART), dim3(FB_TB2), 0, st, (const uint32_t*)pstart,
                               (const uint32_t*)tile_counts, (const uint32_t*)item_idx, (uint32_t)pre->n_level, fb_ntile,
                               (const uint32_t*)fb_long, hist, offs, entries);
        }
    } else if (dm) {
        const uint32_t PW = dm_nwin * dp.ppw;
        {
            StageTimer t(c, ST_PART_COUNT);
            if (!meta_was_clean) HIP_TRY(c, hipMemsetAsync(meta, 0, M_WORDS * 4, st));
            const unsigned dg = (unsigned)((n + BLOCK * DM_DIG_PER - 1) / (BLOCK * DM_DIG_PER));
            if (p.glv)   // d_scalars: the decomposed words (range-checked by k_glv_decompose)
                hipLaunchKernelGGL(k_dm_digits_glv, dim3(dg), dim3(BLOCK), 0, st, d_scalars, (uint32_t)n, dp.n_pad, item_sub);
            else if (dm17)
                hipLaunchKernelGGL(k_dm_digits17, dim3(dg), dim3(BLOCK), 0, st, d_scalars, dp.n, dp.n_pad, item_sub, c->d_flags);
            else
                hipLaunchKernelGGL(k_dm_digits, dim3(dg), dim3(BLOCK), 0, st, d_scalars, dp.n, dp.n_pad, item_sub, c->d_flags);
        }
        {
            StageTimer t(c, ST_PART_SCATTER);
            if (dm17)
                hipLaunchKernelGGL(k_dm_partition<true>, dim3(dp.ntile, dm_nwin), dim3(DM_TB1), 0, st, (const uint16_t*)item_sub, dp, pcount,
                                   tile_counts, item_idx);
            else
                hipLaunchKernelGGL(k_dm_partition<false>, dim3(dp.ntile, dm_nwin), dim3(DM_TB1), 0, st, (const uint16_t*)item_sub, dp, pcount,
                                   tile_counts, item_idx);
                hipLaunchKernelGGL(k_dm_scan, dim3(1), dim3(1024), 0, st, (const uint32_t*)pcount, PW, pstart);
            }
        {
            StageTimer t(c, ST_BUCKET_SORT);
            if (dp.n / dp.ppw <= 4096)
                hipLaunchKernelGGL(k_dm_bucket_sort<16>, dim3(PW), dim3(DM_TB2), 0, st, (const uint32_t*)pstart,
                                   (const uint32_t*)tile_counts, (const uint32_t*)item_idx, dp, p.NB, hist, offs, entries);
            else
                hipLaunchKernelGGL(k_dm_bucket_sort<32>, dim3(PW), dim3(DM_TB2), 0, st, (const uint32_t*)pstart,
                                   (const uint32_t*)tile_counts, (const uint32_t*)item_idx, dp, p.NB, hist, offs, entries);
        }
    } else {
    {
        StageTimer t(c, ST_PART_COUNT);
        if (!meta_was_clean) HIP_TRY(c, hipMemsetAsync(meta, 0, M_WORDS * 4, st));
        hipLaunchKernelGGL(k_part_count, dim3(ntiles), dim3(BLOCK), 0, st, d_scalars, n, p.c, Wd, sp, pcount,
                           c->d_flags, tile_counts);
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(BLOCK), 0, st, pcount, sp.PW, pstart, pcursor);
    }
    if (staged) {
        // LDS-staged sort: keys leave the CU as contiguous runs (n fits the packed item's index field)
        const size_t lds1 = (size_t)(4 * SORT_MAX_PW + STAGE_ITEMS_L1) * 4;
        constexpr int SORT2_TB = 1024;   // level-2 workgroup size (one workgroup per CU: see k_bucket_sort_staged)
        const size_t lds2 = (size_t)(SORT_MAX_SB + SORT2_TB + STAGE_ITEMS) * 4;
        if (!c->staged_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute((const void*)k_part_scatter_staged,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
            HIP_TRY(c, hipFuncSetAttribute((const void*)k_bucket_sort_staged<SORT2_TB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            c->staged_attr_set = true;
        }
        {
            StageTimer t(c, ST_PART_SCATTER);
            if (c->cfg_stage_l1)
                hipLaunchKernelGGL(k_part_scatter_staged, dim3(ntiles), dim3(BLOCK), lds1, st, d_scalars, n, p.c, Wd,
                                   sp, idx_bits, pcursor, item_idx);
            else
                hipLaunchKernelGGL(k_part_scatter_packed, dim3(ntiles), dim3(BLOCK), 0, st, d_scalars, n, p.c, Wd, sp,
                                   idx_bits, pcursor, item_idx, (const uint32_t*)tile_counts);
        }
        {
            StageTimer t(c, ST_BUCKET_SORT);
            hipLaunchKernelGGL(k_bucket_sort_staged<SORT2_TB>, dim3(sp.PW), dim3(SORT2_TB), lds2, st, pstart, item_idx, sp, idx_bits,
                               p.NB, hist, offs, entries);
        }
    } else {
        {
            StageTimer t(c, ST_PART_SCATTER);
            hipLaunchKernelGGL(k_part_scatter, dim3(ntiles), dim3(BLOCK), 0, st, d_scalars, n, p.c, Wd, sp, pcursor,
                               item_idx, item_sub);
        }
        {
            StageTimer t(c, ST_BUCKET_SORT);
            hipLaunchKernelGGL(k_bucket_sort, dim3(sp.PW), dim3(BLOCK), 0, st, pstart, item_idx, item_sub, sp, p.NB,
                               hist, offs, entries);
        }
    }
    }
    // ordering the buckets by length balances the lanes of a wave; with few entries the longest run bounds the kernel
    // either way and the three launches (~25 us) are pure latency
    const bool ordered = nent >= ((size_t)1 << 16);
    if (ordered) {
        StageTimer t(c, ST_ORDER);
        unsigned g = (p.NBT + BLOCK * 8 - 1) / (BLOCK * 8);
        if (g > (unsigned)c->cu_count * 4) g = (unsigned)c->cu_count * 4;
        hipLaunchKernelGGL(k_size_count, dim3(g), dim3(BLOCK), 0, st, hist, p.NBT, bin_count);
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(BLOCK), 0, st, bin_count, (uint32_t)SIZE_BINS, bin_start,
                           bin_cursor);
        hipLaunchKernelGGL(k_size_scatter, dim3(g), dim3(BLOCK), 0, st, hist, p.NBT, bin_cursor, order);
    }
    if (c->deferred_tail) {   // the previous MSM's tail goes out now, behind this MSM's sort
        HIP_TRY(c, hipEventRecord(c->ev_sortdone, st));
        TRY(flush_deferred_tail(c, true));
    }
    if (piped) {   // from here on: the accumulation stream
        HIP_TRY(c, hipEventRecord(c->ev_sorted[sq], st));
        HIP_TRY(c, hipStreamWaitEvent(c->acc_stream, c->ev_sorted[sq], 0));
        st = c->acc_stream;
    }
    if (c->tail_pending[par]) {  // the tail TAIL_SLOTS MSMs ago read this slot's buckets / segsum / wsum
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_tail[par], 0));
        c->tail_pending[par] = false;
    }
    // lanes per bucket: keep >= ~8192 waves in flight (3 per SIMD x 1024 SIMDs, 2-3 rounds) when buckets are few
    // With few buckets the kernel is bound by its longest run (one mixed add is ~6 us of dependent latency): 8 lanes
    // per bucket turn a 60-entry run into 8 entries + 7 adds of the combine.
    uint32_t lpb = 1;
    if (c->cfg_lpb) lpb = (uint32_t)c->cfg_lpb;
    else   // ... but only while the mean run still covers the slices (measured, profiles/r01_sweeps.txt): sparse buckets
           // (small MSMs with wide windows) gain nothing from slices and pay lpb - 1 additions per bucket in the combine
        while (lpb < 8 && (size_t)p.NBT * lpb < (size_t)8192 * 64 &&
               (chain ? c->chain_n * (size_t)Wd * (p.glv ? 2 : 1) : nent) >= (size_t)lpb * p.NBT)
            lpb *= 2;   // (a chain's slices all cut their buckets alike: the slots are resumed)
    if (c->bases_hook) {
        TRY(c->bases_hook(st));
        if (endo_late)
            hipLaunchKernelGGL(k_bases_endo_x, dim3(grid_for(c, n_base)), dim3(BLOCK), 0, st, d_bases, n_base, (uint8_t*)c->endo_buf.p);
    }
    uint8_t* acc_out = buckets;
    if (lpb > 1) {
        TRY(ensure(c, c->parts, (size_t)p.NBT * lpb * XYZZ_BYTES));
        acc_out = (uint8_t*)c->parts.p;
    }
    // (experiment knob: 256-thread workgroups + H2AGG_ACC_LDS pin the accumulation at exactly N waves per SIMD and leave the rest
    // of the CU — registers and LDS — to whatever else is in flight; see profiles/r03_sweeps.txt section 10)
    static const int acc_block = knob("H2AGG_ACC_BLOCK") ? atoi(knob("H2AGG_ACC_BLOCK")) : 64;
    // the generic kernel (compiler-scheduled formulas, 166 VGPRs) lives in the measure build only, behind the debug key
    // lean_acc = 0 / H2AGG_ACC=generic, for A/B runs; the shipped library carries the lean one alone
#ifdef H2AGG_MEASURE_KNOBS
    static const bool lean_env = !(knob("H2AGG_ACC") && !strcmp(knob("H2AGG_ACC"), "generic"));
    const bool lean = lean_env && c->dbg_lean_acc;
#else
    const bool lean = true;
#endif
    uint32_t* fix_list = nullptr;
    if (lean) {
        TRY(ensure(c, c->fix_list[sq], (size_t)p.NBT * lpb * 8));
        fix_list = (uint32_t*)c->fix_list[sq].p;
    }
    debug_sync(2);
    chaos_wait(2, st);
    {
        StageTimer t(c, ST_ACCUM, st);
        // one-wave workgroups: a 4-wave workgroup needs a free slot on all four SIMDs of a CU at once and its waves retire
        // at different times; single waves fill any slot as it frees up (2^20 points: 1.74 -> 1.67 ms/step)
        static const int acc_lds = knob("H2AGG_ACC_LDS") ? atoi(knob("H2AGG_ACC_LDS")) : 0;   // experiment: unused LDS per wave caps the occupancy
        if (lean) {   // 128 VGPRs, four waves per SIMD; exceptional cases go to fix_list (msm_kernels.hpp)
#ifdef H2AGG_MEASURE_KNOBS   // (one chain per product instead of two in lock step: an A/B variant, not in the shipped library)
            static const bool lean_dual = !(knob("H2AGG_ACC") && !strcmp(knob("H2AGG_ACC"), "lean1"));
            auto kacc = lean_dual ? (chain == CHAIN_FIRST ? k_msm_accumulate_lean<1, true> : (chain == CHAIN_MID || chain == CHAIN_LAST) ? k_msm_accumulate_lean<2, true> : k_msm_accumulate_lean<0, true>)
                                  : (chain == CHAIN_FIRST ? k_msm_accumulate_lean<1, false> : (chain == CHAIN_MID || chain == CHAIN_LAST) ? k_msm_accumulate_lean<2, false> : k_msm_accumulate_lean<0, false>);
#else
            auto kacc = chain == CHAIN_FIRST ? k_msm_accumulate_lean<1, true> : (chain == CHAIN_MID || chain == CHAIN_LAST) ? k_msm_accumulate_lean<2, true> : k_msm_accumulate_lean<0, true>;
#endif
            hipLaunchKernelGGL(kacc, dim3((unsigned)(((size_t)p.NBT * lpb + 63) / 64)), dim3(64), (size_t)acc_lds,
                               st, d_bases, d_endo_x, entries, offs, hist, ordered ? order : (uint32_t*)nullptr, p.NBT, p.big, lpb, acc_out,
                               big_list, big_keys, big_count, fix_list);
        } else {
#ifdef H2AGG_MEASURE_KNOBS
        auto kacc = chain == CHAIN_FIRST ? k_msm_accumulate<1> : (chain == CHAIN_MID || chain == CHAIN_LAST) ? k_msm_accumulate<2> : k_msm_accumulate<0>;
        hipLaunchKernelGGL(kacc, dim3((unsigned)(((size_t)p.NBT * lpb + acc_block - 1) / acc_block)), dim3(acc_block), (size_t)acc_lds,
                           st, d_bases, d_endo_x, entries, offs, hist, ordered ? order : (uint32_t*)nullptr, p.NBT, p.big, lpb, acc_out,
                           big_list, big_keys, big_count);
#endif
        }
    }
    // Buckets longer than `big` (skewed scalars; none for uniform ones, where the two launches below only find empty lists):
    // with alternating sort outputs they leave the bulk stream and go in front of the bucket reduction on the tail stream,
    // followed by the zeroing of this slot's counters for the MSM after next.
    const bool tail_big = altbuf && c->overlap_level >= 2 && lpb == 1 && (chain == CHAIN_OFF || chain == CHAIN_LAST) && !split;
    const uint32_t resume = (chain == CHAIN_MID || chain == CHAIN_LAST) ? 1u : 0u;
    const bool chain_open = chain == CHAIN_FIRST || chain == CHAIN_MID;   // no tail yet
    // (no bucket can hold more keys than its bucket set receives: a multi_exp of a dozen points skips the two launches,
    // ~35 us at the head of its tail)
    const bool big_possible = nent / WT > p.big;
    auto big_kernels = [=](hipStream_t bs) {
        StageTimer t(c, ST_ACCUM_BIG, bs);
        if (lean)   // (grid-stride over a list that is empty but for the one-limb filter's false positives and adversarial inputs)
            hipLaunchKernelGGL(k_msm_accumulate_fix, dim3((unsigned)std::min<size_t>((size_t)c->cu_count * 12, ((size_t)p.NBT * lpb + 63) / 64)), dim3(64), 0, bs, d_bases, d_endo_x, entries, offs,
                               hist, lpb, acc_out, (const uint32_t*)big_count, (const uint32_t*)fix_list);
        if (!big_possible && lpb == 1) return;
        size_t grid = max_slots;
        const size_t cap = (size_t)c->cu_count * 8;   // one-wave workgroups, grid-stride over the (usually empty) lists
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(k_msm_accumulate_big, dim3((unsigned)grid), dim3(64), 0, bs, d_bases, d_endo_x, entries, offs, hist,
                           acc_out, lpb, big_part, big_list, big_count, resume);
        size_t gk = max_keys < cap ? max_keys : cap;
        hipLaunchKernelGGL(k_msm_big_combine, dim3((unsigned)gk), dim3(64), 0, bs, big_part, big_keys, big_count,
                           acc_out, lpb, resume);
        // (in a chain only the last slice folds the slice slots, and it folds ALL of them: a bucket that is over-long in this
        // slice may hold ordinary partial sums from earlier ones)
        if (lpb > 1 && !chain_open) {
            const uint32_t cbig = chain ? 0xffffffffu : p.big;
            const unsigned gq = (unsigned)(((size_t)p.NBT * 2 * lpb + BLOCK - 1) / BLOCK);
            if (p.NBT > 16384u || (lpb != 2 && lpb != 4 && lpb != 8))   // many buckets: one lane each (work, not latency)
                hipLaunchKernelGGL(k_msm_bucket_combine, dim3((p.NBT + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, bs,
                                   (const uint8_t*)acc_out, hist, p.NBT, cbig, lpb, buckets);
            else if (lpb == 8)
                hipLaunchKernelGGL(k_msm_bucket_combine_par4<8>, dim3(gq), dim3(BLOCK), 0, bs, (const uint8_t*)acc_out, hist, p.NBT, cbig, buckets);
            else if (lpb == 4)
                hipLaunchKernelGGL(k_msm_bucket_combine_par4<4>, dim3(gq), dim3(BLOCK), 0, bs, (const uint8_t*)acc_out, hist, p.NBT, cbig, buckets);
            else
                hipLaunchKernelGGL(k_msm_bucket_combine_par4<2>, dim3(gq), dim3(BLOCK), 0, bs, (const uint8_t*)acc_out, hist, p.NBT, cbig, buckets);
        }
    };
    if (!tail_big) {
        big_kernels(st);
        if (altbuf) {
            HIP_TRY(c, hipEventRecord(c->ev_accdone[sq], st));
            c->accdone_pending[sq] = true;
        }
    }
    if (altbuf) c->sort_par ^= 1;
    if (chain_open) {   // the bucket set stays open for the next slice: no tail, the slot does not rotate
        HIP_TRY(c, hipGetLastError());
        profile_end_call(c);
        return H2AGG_OK;
    }
    // Everything after the bucket accumulation is latency-shaped (one wave per SIMD or less): bucket
    // reduction, per-window sums, Horner tail.  In overlap mode it runs on one of the context's tail streams, under the
    // accumulation of the next MSM (launched from that MSM's call, behind its sort: `deferred_tail`); results are picked
    // up by join_tails().
#ifdef H2AGG_MEASURE_KNOBS   // timing experiments only (WRONG results): 1 skips the bucket reduction, 2 the window sums, 4 the Horner tail
    static const int dbg_skip = knob("H2AGG_DBG_SKIP") ? atoi(knob("H2AGG_DBG_SKIP")) : 0;
#else
    constexpr int dbg_skip = 0;   // (a shipped library has no switch that changes results)
#endif
    // two-dimensional bucket reduction for 16-bit windows (msm_kernels.hpp); H2AGG_REDUCE=segments keeps the segment kernels
    static const bool r2d_env_off = knob("H2AGG_REDUCE") && !strcmp(knob("H2AGG_REDUCE"), "segments");
    const int r2d_lc = p.NB == (uint32_t)(R2D_ROWS * R2D<7>::COLS) ? 7 : p.NB == (uint32_t)(R2D_ROWS * R2D<8>::COLS) ? 8 : 0;
    const bool r2d = !r2d_env_off && !c->cfg_seg && !pre && r2d_lc != 0;
    uint32_t* ticket = nullptr;
    if (r2d || fbdm) {
        DevBuf& tk = c->r2d_ticket[par];   // arrival counters of the two half-window workgroups: zero between MSMs
        const size_t tickets = fbdm ? (size_t)FB_R2D_WINDOWS : (size_t)WT;
        if (tickets * 4 > tk.cap) {
            TRY(ensure(c, tk, tickets * 4));
            HIP_TRY(c, hipMemset(tk.p, 0, tk.cap));
        }
        ticket = (uint32_t*)tk.p;
    }
    debug_sync(4);
    const bool tails_off_stream = c->tail_overlap && c->overlap_level >= 2 && !split;
    const bool final_off_stream = c->tail_overlap && !split;
    uint8_t* const res_xyzz = c->d_res_xyzz;
    if (tails_off_stream) HIP_TRY(c, hipEventRecord(c->ev_bulk[par], st));
    // `from`: the stream the accumulation ran on.  behind_sort: the tail stream also waits for c->ev_sortdone.
    auto tail_fn = [=](bool behind_sort) -> int {
        hipStream_t ts = st;
        if (tails_off_stream) {
            ts = c->tail_streams[par];
            HIP_TRY(c, hipStreamWaitEvent(ts, c->ev_bulk[par], 0));
            if (behind_sort) HIP_TRY(c, hipStreamWaitEvent(ts, c->ev_sortdone, 0));
        }
        chaos_wait(1, ts);
        if (tail_big) {
            big_kernels(ts);
            HIP_TRY(c, hipMemsetAsync(meta, 0, M_WORDS * 4, ts));
            c->meta_clean[sq] = true;
            HIP_TRY(c, hipEventRecord(c->ev_accdone[sq], ts));
            c->accdone_pending[sq] = true;
        }
        if (fbdm) {   // one set of 2^19 buckets as 16 grids of 256 x 128 (msm_kernels.hpp, above k_fb_fold)
            constexpr uint32_t per_w = (uint32_t)R2D<7>::THREADS, total = FB_R2D_WINDOWS * per_w;
            uint8_t* halves = segsum + XYZZ_BYTES * (size_t)total;
            uint8_t* tsum = halves + XYZZ_BYTES * (size_t)(2 * FB_R2D_WINDOWS);
            uint8_t* w2 = wsum + XYZZ_BYTES * (size_t)FB_R2D_WINDOWS;
            {
                StageTimer t(c, ST_REDUCE, ts);
                hipLaunchKernelGGL(k_fb_fold, dim3((FB_XB >> FB_XPARTS_LOG) / 64), dim3(64), 0, ts, buckets);
                hipLaunchKernelGGL(k_msm_reduce2d_parts<7>, dim3(total / 64), dim3(64), 0, ts, (const uint8_t*)buckets, total, segsum);
            }
            {
                StageTimer t(c, ST_WINDOW_SUM, ts);
                hipLaunchKernelGGL(k_msm_reduce2d_window<7>, dim3(FB_R2D_WINDOWS, 2), dim3(R2D_TB), 0, ts, (const uint8_t*)segsum, halves, ticket,
                                   wsum, tsum);
            }
            if (final_off_stream && !tails_off_stream) {   // overlap level 1: only the last chain leaves the stream
                HIP_TRY(c, hipEventRecord(c->ev_bulk[par], st));
                HIP_TRY(c, hipStreamWaitEvent(c->tail_streams[par], c->ev_bulk[par], 0));
                ts = c->tail_streams[par];
            }
            {
                StageTimer t(c, ST_FINAL, ts);
                hipLaunchKernelGGL(k_fb_wsum, dim3(1), dim3(64), 0, ts, (const uint8_t*)wsum, (const uint8_t*)tsum, w2);
                hipLaunchKernelGGL(k_msm_final_lp, dim3(1), dim3(64), 0, ts, (const uint8_t*)w2, 15, 2, res_xyzz, d_out_jac);
            }
            if (final_off_stream) {
                HIP_TRY(c, hipEventRecord(c->ev_tail[par], c->tail_streams[par]));
                c->tail_pending[par] = true;
            }
            return H2AGG_OK;
        }
        if (r2d) {
            const uint32_t per_w = r2d_lc == 7 ? (uint32_t)R2D<7>::THREADS : (uint32_t)R2D<8>::THREADS;
            if (!(dbg_skip & 1)) {
                StageTimer t(c, ST_REDUCE, ts);
                const uint32_t total = WT * per_w;
                if (r2d_lc == 7)
                    hipLaunchKernelGGL(k_msm_reduce2d_parts<7>, dim3(total / 64), dim3(64), 0, ts, (const uint8_t*)buckets, total, segsum);
                else
                    hipLaunchKernelGGL(k_msm_reduce2d_parts<8>, dim3(total / 64), dim3(64), 0, ts, (const uint8_t*)buckets, total, segsum);
            }
            if (!(dbg_skip & 2)) {
                StageTimer t(c, ST_WINDOW_SUM, ts);
                if (r2d_lc == 7)
                    hipLaunchKernelGGL(k_msm_reduce2d_window<7>, dim3(WT, 2), dim3(R2D_TB), 0, ts, (const uint8_t*)segsum,
                                       segsum + XYZZ_BYTES * (size_t)WT * per_w, ticket, wsum);
                else
                    hipLaunchKernelGGL(k_msm_reduce2d_window<8>, dim3(WT, 2), dim3(R2D_TB), 0, ts, (const uint8_t*)segsum,
                                       segsum + XYZZ_BYTES * (size_t)WT * per_w, ticket, wsum);
            }
        } else {
            if (!(dbg_skip & 1)) {
                StageTimer t(c, ST_REDUCE, ts);
                if (par4)
                    hipLaunchKernelGGL(k_msm_reduce_segments_par4, dim3((4 * nseg_total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ts,
                                       buckets, p.NB, p.seg, p.spw, nseg_total, segsum);
                else
                    hipLaunchKernelGGL(k_msm_reduce_segments, dim3((nseg_total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ts, buckets,
                                       p.NB, p.seg, p.spw, nseg_total, segsum);
            }
            if (!(dbg_skip & 2)) {
                StageTimer t(c, ST_WINDOW_SUM, ts);
                if (par4)
                    hipLaunchKernelGGL(k_msm_window_sum_par4, dim3(WT), dim3(PAR4_THREADS), 0, ts, segsum, p.spw, wsum);
                else
                    hipLaunchKernelGGL(k_msm_window_sum, dim3(WT), dim3(BLOCK), 0, ts, segsum, p.spw, wsum);
            }
        }
        if (final_off_stream && !tails_off_stream) {   // overlap level 1: only the Horner tail leaves the stream
            HIP_TRY(c, hipEventRecord(c->ev_bulk[par], st));
            HIP_TRY(c, hipStreamWaitEvent(c->tail_streams[par], c->ev_bulk[par], 0));
            ts = c->tail_streams[par];
        }
        if (!(dbg_skip & 4)) {
            StageTimer t(c, ST_FINAL, ts);
            // the Horner chain in the limb-parallel form (lp_kernels.hpp): 1.2 instead of 2.0 us per doubling
            static const bool final_par4 = knob("H2AGG_FINAL") && !strcmp(knob("H2AGG_FINAL"), "par4");
            if (final_par4)
                hipLaunchKernelGGL(k_msm_final, dim3(batch), dim3(64), 0, ts, wsum, p.c, W1, (batch > 1 && !split) ? (uint8_t*)nullptr : res_xyzz,
                                   d_out_jac);
            else
                hipLaunchKernelGGL(k_msm_final_lp, dim3(batch), dim3(64), 0, ts, wsum, p.c, W1, (batch > 1 && !split) ? (uint8_t*)nullptr : res_xyzz,
                                   d_out_jac);
        }
        if (final_off_stream) {
            HIP_TRY(c, hipEventRecord(c->ev_tail[par], c->tail_streams[par]));
            c->tail_pending[par] = true;
        }
        return H2AGG_OK;
    };
    // deferral needs another MSM to carry it; a full per-stage profiling pass keeps every stage inside its own call
    static const bool defer_env_off = knob("H2AGG_DEFER_TAILS") && !strcmp(knob("H2AGG_DEFER_TAILS"), "0");
    // (and it only pays from 2^20 points on — measured, profiles/r02_sweeps.txt: below that the tail is a large share of the
    // MSM and wants to start at once; the two multi_exps of an evaluation in particular)
    // (not for the fixed-base levels of big tables: their sort's whole-CU workgroups collide with a tail either way — a tail always
    // ends up beside some later MSM's sort, profiles/r05_sweeps.txt section 1(g) — and the batch of 16 x 2^22 measures 71.6 ms with
    // the tail launched at once against 72.5 deferred)
    if (tails_off_stream && !defer_env_off && !(c->profiling && c->prof_only < 0) && n >= ((size_t)1 << 20) && !fbdm) {
        c->deferred_tail = tail_fn;
    } else {
        TRY(tail_fn(false));
    }
    // the slot (buckets / segsum / wsum / XYZZ result) rotates on every MSM, overlap or not: a caller may queue
    // two MSMs and read both results afterwards (evaluate_multiopen_proof does)
    c->parity = (c->parity + 1) % h2agg_ctx::TAIL_SLOTS;
    debug_sync(8);
    HIP_TRY(c, hipGetLastError());
    profile_end_call(c);
    return H2AGG_OK;
}

int set_identity_jac(uint8_t out[96]) {
    memset(out, 0, 96);
    out[32] = 1;
    return 0;
}

int fetch_result_jac(h2agg_ctx* c, uint8_t out[96]) {
    TRY(join_tails(c));
    HIP_TRY(c, hipMemcpyAsync(c->h_pinned, c->d_res_jac, 96, hipMemcpyDeviceToHost, c->stream));
    TRY(finish(c));
    memcpy(out, c->h_pinned, 96);
    return H2AGG_OK;
}

int bind(h2agg_ctx* c) {
    crumb((uintptr_t)__builtin_return_address(0), 0);
    if (!c) return H2AGG_ERR_INVALID;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, H2AGG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return H2AGG_OK;
}

}  // namespace

extern "C" {

int h2agg_create(int device_ordinal, h2agg_ctx** out) try {
    if (!out) return H2AGG_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (device_ordinal < 0 || hipGetDeviceCount(&count) != hipSuccess || device_ordinal >= count)
        return H2AGG_ERR_HIP;  // no CPU mode: a HIP device is required
    if (hipSetDevice(device_ordinal) != hipSuccess) return H2AGG_ERR_HIP;
    h2agg_ctx* c = new h2agg_ctx();
    c->device = device_ordinal;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) {
        delete c;
        return H2AGG_ERR_HIP;
    }
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    char buf[256];
    snprintf(buf, sizeof buf, "h2agg 0.1 %s cu=%d", prop.gcnArchName, c->cu_count);
    c->desc = buf;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&c->h_pinned, 4096) != hipSuccess || ensure(c, c->small, 4096) != H2AGG_OK) {
        h2agg_destroy(c);
        return H2AGG_ERR_HIP;
    }
    c->stream = c->own_stream;
    for (int k = 0; k < h2agg_ctx::TAIL_SLOTS; ++k) {
        // H2AGG_TAIL_PRIO (experiment): -1 = lowest, 1 = highest stream priority for the tail streams
        static const int tail_prio = knob("H2AGG_TAIL_PRIO") ? atoi(knob("H2AGG_TAIL_PRIO")) : 0;
        int pr_lo = 0, pr_hi = 0;
        hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);   // lo = least (numerically greatest)
        if ((tail_prio ? hipStreamCreateWithPriority(&c->tail_streams[k], hipStreamNonBlocking, tail_prio < 0 ? pr_lo : pr_hi)
                       : hipStreamCreateWithFlags(&c->tail_streams[k], hipStreamNonBlocking)) != hipSuccess) {
            h2agg_destroy(c);
            return H2AGG_ERR_HIP;
        }
        hipEventCreateWithFlags(&c->ev_bulk[k], EV_SYNC_FLAGS);
        hipEventCreateWithFlags(&c->ev_tail[k], EV_SYNC_FLAGS);
    }
    int apr_lo = 0, apr_hi = 0;
    hipDeviceGetStreamPriorityRange(&apr_lo, &apr_hi);
    static const int acc_prio = knob("H2AGG_ACC_PRIO") ? atoi(knob("H2AGG_ACC_PRIO")) : 0;   // experiment: -1 lowest
    if ((acc_prio ? hipStreamCreateWithPriority(&c->acc_stream, hipStreamNonBlocking, acc_prio < 0 ? apr_lo : apr_hi)
                  : hipStreamCreateWithFlags(&c->acc_stream, hipStreamNonBlocking)) != hipSuccess) {
        h2agg_destroy(c);
        return H2AGG_ERR_HIP;
    }
    // The auxiliary and the copy stream are made HERE, in a fixed order behind the others, not at their first use: where the
    // runtime places a stream depends on how many streams the process has made before it (profiles/r03_sweeps.txt section 18:
    // back-to-back MSMs run at 1.25 or at 1.55 ms per step depending on the number of streams made between the main stream and
    // the tail streams), and a placement that changes with what else the process did first is not reproducible.
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        h2agg_destroy(c);
        return H2AGG_ERR_HIP;
    }
    for (int k = 0; k < 4; ++k)
        if (hipStreamCreateWithFlags(&c->spare_streams[k], hipStreamNonBlocking) != hipSuccess) {
            h2agg_destroy(c);
            return H2AGG_ERR_HIP;
        }
    hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_aux_go, hipEventDisableTiming);
    for (int k = 0; k < MSM_MAX_SLICES; ++k) {
        hipEventCreateWithFlags(&c->ev_copy[k], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev_copy_s[k], hipEventDisableTiming);
    }
    hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_sortdone, EV_SYNC_FLAGS);
    for (int q = 0; q < 2; ++q) {
        hipEventCreateWithFlags(&c->ev_sorted[q], EV_SYNC_FLAGS);
        hipEventCreateWithFlags(&c->ev_accdone[q], EV_SYNC_FLAGS);
    }
    c->d_flags = (uint32_t*)c->small.p;
    c->d_res_xyzz = (uint8_t*)c->small.p + 1024;   // 144 B
    c->d_res_jac = (uint8_t*)c->small.p + 256;   // 96 B
    hipMemset(c->small.p, 0, 4096);
    if (place_streams(c) != H2AGG_OK) {
        h2agg_destroy(c);
        return H2AGG_ERR_HIP;
    }
    *out = c;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

}  // extern "C"
namespace {
void comm_release(h2agg_ctx* c);   // csrc/comm.inc
void agg_plans_release(h2agg_ctx* c);   // csrc/verifier.inc
}
extern "C" {
void h2agg_destroy(h2agg_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    comm_release(c);
    agg_plans_release(c);
    flush_deferred_tail(c, false);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->acc_stream) hipStreamSynchronize(c->acc_stream);
    for (int k = 0; k < h2agg_ctx::TAIL_SLOTS; ++k)
        if (c->tail_streams[k]) hipStreamSynchronize(c->tail_streams[k]);
    static_assert(h2agg_ctx::TAIL_SLOTS == 3, "the list below names every tail slot's buffers");
    DevBuf* bufs[] = {&c->in_a,  &c->in_b,     &c->in_c,     &c->out,   &c->tmp_bases, &c->comb, &c->psd_spec, &c->tr_in, &c->tr_points, &c->tr_elems, &c->tr_chal, &c->inst_vals, &c->inst_jac, &c->inst_aff, &c->agg_elems, &c->hist[0], &c->hist[1],
                      &c->offs[0], &c->offs[1], &c->pmeta[0], &c->pmeta[1], &c->item_idx, &c->item_sub, &c->order[0], &c->order[1], &c->entries[0], &c->entries[1],
                      &c->r2d_ticket[0], &c->r2d_ticket[1], &c->r2d_ticket[2], &c->buckets[0], &c->buckets[1], &c->buckets[2], &c->segsum[0], &c->segsum[1], &c->segsum[2],
                      &c->wsum[0], &c->wsum[1], &c->wsum[2], &c->big_list[0], &c->big_list[1], &c->big_keys[0], &c->big_keys[1], &c->big_part[0], &c->big_part[1], &c->fix_list[0], &c->fix_list[1], &c->glv_buf, &c->parts, &c->small, &c->endo_buf, &c->tile_counts, &c->fb_long,
                      &c->sch_regs, &c->sch_in, &c->sch_scalars[0], &c->sch_scalars[1], &c->sch_bases[0], &c->sch_bases[1], &c->sch_endo};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    for (auto& kv : c->tables) {
        if (kv.second.d) hipFree(kv.second.d);
        if (kv.second.endo_x) hipFree(kv.second.endo_x);
        kv.second.free_combs();
        if (kv.second.pre) hipFree(kv.second.pre);
    }
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_down) hipHostFree(c->h_down);
    if (c->copy_stream) {
        hipStreamSynchronize(c->copy_stream);
        for (int k = 0; k < MSM_MAX_SLICES; ++k) {
            if (c->ev_copy[k]) hipEventDestroy(c->ev_copy[k]);
            if (c->ev_copy_s[k]) hipEventDestroy(c->ev_copy_s[k]);
        }
        if (c->ev_ready) hipEventDestroy(c->ev_ready);
        hipStreamDestroy(c->copy_stream);
    }
    if (c->aux_stream) {
        hipStreamSynchronize(c->aux_stream);
        if (c->ev_aux) hipEventDestroy(c->ev_aux);
        if (c->ev_aux_go) hipEventDestroy(c->ev_aux_go);
        hipStreamDestroy(c->aux_stream);
    }
    for (int r = 0; r < h2agg_ctx::PROF_RING; ++r)
        for (int s = 0; s < ST_N; ++s)
            for (int k = 0; k < 2; ++k)
                if (c->prof[r].ev[s][k]) hipEventDestroy(c->prof[r].ev[s][k]);
    for (int k = 0; k < h2agg_ctx::TAIL_SLOTS; ++k) {
        if (c->ev_bulk[k]) hipEventDestroy(c->ev_bulk[k]);
        if (c->ev_tail[k]) hipEventDestroy(c->ev_tail[k]);
        if (c->tail_streams[k]) hipStreamDestroy(c->tail_streams[k]);
    }
    if (c->ev_sortdone) hipEventDestroy(c->ev_sortdone);
    for (int q = 0; q < 2; ++q) {
        if (c->ev_sorted[q]) hipEventDestroy(c->ev_sorted[q]);
        if (c->ev_accdone[q]) hipEventDestroy(c->ev_accdone[q]);
    }
    if (c->acc_stream) hipStreamDestroy(c->acc_stream);
    for (int k = 0; k < 4; ++k)
        if (c->spare_streams[k]) hipStreamDestroy(c->spare_streams[k]);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

const char* h2agg_last_error(const h2agg_ctx* c) { return c ? c->err.c_str() : "null context"; }
const char* h2agg_describe(h2agg_ctx* c) { return c ? c->desc.c_str() : ""; }

int h2agg_set_stream(h2agg_ctx* c, void* hip_stream) try {
    TRY(bind(c));
    TRY(join_tails(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const hipStream_t was = c->stream;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (c->stream != was) {   // another main stream, another queue: hand the roles out again
        // The probe launches kernels on the caller's stream and waits for them: not possible while that stream is being
        // captured into a graph — the roles then stay as they are (H2AGG_NO_PLACE skips the probe altogether).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hip_stream && hipStreamIsCapturing(c->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return H2AGG_OK;
        (void)hipGetLastError();
        HIP_TRY(c, hipDeviceSynchronize());
        TRY(place_streams(c));
    }
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_synchronize(h2agg_ctx* c) try {
    TRY(bind(c));
    TRY(join_tails(c));
    // also the place where status flags raised by asynchronous calls surface (e.g. a scalar >= r handed to
    // h2agg_g1_msm_device_async -> H2AGG_ERR_NONCANONICAL here)
    TRY(clear_flags(c));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ---------------------------------------------------------------- Fr
int h2agg_fr_batch_op(h2agg_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) try {
    TRY(bind(c));
    if (op < H2AGG_OP_ADD || op > H2AGG_OP_DIV) return fail(c, H2AGG_ERR_INVALID, "unknown field op");
    if (n == 0) return H2AGG_OK;
    const bool binary = op <= H2AGG_OP_MUL || op == H2AGG_OP_DIV;
    if (!a || !out || (binary && !b)) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 32 * n));
    TRY(ensure(c, c->out, 32 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, a, 32 * n, hipMemcpyHostToDevice, c->stream));
    if (binary) {
        TRY(ensure(c, c->in_b, 32 * n));
        HIP_TRY(c, hipMemcpyAsync(c->in_b.p, b, 32 * n, hipMemcpyHostToDevice, c->stream));
    }
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_fr_batch_op, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, op, (const uint8_t*)c->in_a.p,
                       (const uint8_t*)c->in_b.p, n, (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_fr_batch_pow_constant(h2agg_ctx* c, const uint8_t* a, size_t n, uint64_t exponent, uint8_t* out) try {
    TRY(bind(c));
    if (exponent < 1) return fail(c, H2AGG_ERR_INVALID, "assert!(exponent >= 1) failed (arith/field.rs:89)");
    if (n == 0) return H2AGG_OK;
    if (!a || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 32 * n));
    TRY(ensure(c, c->out, 32 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, a, 32 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_fr_batch_pow, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, n,
                       exponent, (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_fr_mul_add_accumulate(h2agg_ctx* c, const uint8_t* v, size_t n, const uint8_t b[32], uint8_t out[32]) try {
    TRY(bind(c));
    if ((n && !v) || !b || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 32 * n + 32));
    TRY(ensure(c, c->in_b, 32));
    TRY(ensure(c, c->out, 32));
    if (n) HIP_TRY(c, hipMemcpyAsync(c->in_a.p, v, 32 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_b.p, b, 32, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_fr_horner, dim3(1), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, n,
                       (const uint8_t*)c->in_b.p, (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 32, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_fr_sum_with_coeff_and_constant(h2agg_ctx* c, const uint8_t* x, const uint8_t* coeff, size_t n,
                                         const uint8_t b[32], uint8_t out[32]) try {
    TRY(bind(c));
    if ((n && (!x || !coeff)) || !b || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 32 * n + 32));
    TRY(ensure(c, c->in_b, 32 * n + 32));
    TRY(ensure(c, c->in_c, 32));
    TRY(ensure(c, c->out, 32));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(c->in_a.p, x, 32 * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->in_b.p, coeff, 32 * n, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(c->in_c.p, b, 32, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_fr_sum_coeff, dim3(1), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p,
                       (const uint8_t*)c->in_b.p, n, (const uint8_t*)c->in_c.p, (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 32, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ---------------------------------------------------------------- G1 batch
int h2agg_g1_batch_add(h2agg_ctx* c, const uint8_t* a, const uint8_t* b, size_t n, int subtract, uint8_t* out) try {
    TRY(bind(c));
    if (n == 0) return H2AGG_OK;
    if (!a || !b || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 96 * n));
    TRY(ensure(c, c->in_b, 96 * n));
    TRY(ensure(c, c->out, 96 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, a, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_b.p, b, 96 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_g1_batch_add, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p,
                       (const uint8_t*)c->in_b.p, n, subtract, (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 96 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_batch_scalar_mul(h2agg_ctx* c, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out) try {
    TRY(bind(c));
    if (n == 0) return H2AGG_OK;
    if (!bases || !scalars || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 64 * n));
    TRY(ensure(c, c->in_b, 32 * n));
    TRY(ensure(c, c->out, 96 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, bases, 64 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_b.p, scalars, 32 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    // GLV + signed window-4 ladder, four lanes per point (csrc/scalar_mul_kernels.hpp); H2AGG_SCALAR_MUL=ladder selects the
    // round-1 bit-serial kernel for A/B measurements
    static const bool ladder = knob("H2AGG_SCALAR_MUL") && !strcmp(knob("H2AGG_SCALAR_MUL"), "ladder");
    if (ladder) {
        hipLaunchKernelGGL(k_g1_batch_scalar_mul, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream,
                           (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_b.p, n, (uint8_t*)c->out.p, c->d_flags);
    } else {
        size_t groups = (n + SM_GROUPS - 1) / SM_GROUPS;
        const size_t cap = (size_t)c->cu_count * 16;
        if (groups > cap) groups = cap;
        hipLaunchKernelGGL(k_g1_batch_scalar_mul_w4, dim3((unsigned)groups), dim3(SM_THREADS), 0, c->stream,
                           (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_b.p, n, (uint8_t*)c->out.p, c->d_flags);
    }
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 96 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_batch_to_affine(h2agg_ctx* c, const uint8_t* in, size_t n, uint8_t* out) try {
    TRY(bind(c));
    if (n == 0) return H2AGG_OK;
    if (!in || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 96 * n));
    TRY(ensure(c, c->out, 64 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, in, 96 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_g1_batch_to_affine, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream,
                       (const uint8_t*)c->in_a.p, n, (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 64 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_batch_to_affine_device(h2agg_ctx* c, const uint8_t* d_in_jac, size_t n, uint8_t* out) try {
    TRY(bind(c));
    if (n == 0) return H2AGG_OK;
    if (!d_in_jac || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(join_tails(c));   // the inputs are typically results of h2agg_g1_msm_device_async
    TRY(ensure(c, c->out, 64 * n));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_g1_batch_to_affine, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, d_in_jac, n,
                       (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 64 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_batch_decompress(h2agg_ctx* c, const uint8_t* in, size_t n, uint8_t* out_aff, uint8_t* ok) try {
    TRY(bind(c));
    if (n == 0) return H2AGG_OK;
    if (!in || !out_aff) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 32 * n));
    TRY(ensure(c, c->out, 64 * n));
    TRY(ensure(c, c->in_b, n + 16));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, in, 32 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_g1_batch_decompress, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, n,
                       (uint8_t*)c->out.p, (uint8_t*)c->in_b.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out_aff, c->out.p, 64 * n, hipMemcpyDeviceToHost, c->stream));
    if (ok) HIP_TRY(c, hipMemcpyAsync(ok, c->in_b.p, n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}
int h2agg_g1_batch_compress(h2agg_ctx* c, const uint8_t* aff, size_t n, uint8_t* out) try {
    TRY(bind(c));
    if (n == 0) return H2AGG_OK;
    if (!aff || !out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 64 * n));
    TRY(ensure(c, c->out, 32 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, aff, 64 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_g1_batch_compress, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, n,
                       (uint8_t*)c->out.p, c->d_flags);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_sum(h2agg_ctx* c, const uint8_t* in, size_t n, uint8_t out[96]) try {
    TRY(bind(c));
    if (!out || (n && !in)) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, 96 * n + 96));
    if (n) HIP_TRY(c, hipMemcpyAsync(c->in_a.p, in, 96 * n, hipMemcpyHostToDevice, c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_g1_sum, dim3(1), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, n, c->d_res_jac,
                       c->d_flags);
    return fetch_result_jac(c, out);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ---------------------------------------------------------------- base tables
int h2agg_bases_upload(h2agg_ctx* c, const uint8_t* bases, size_t n, uint64_t* handle_out) try {
    TRY(bind(c));
    if (!bases || !handle_out || n == 0) return fail(c, H2AGG_ERR_INVALID, "null buffer or n == 0");
    Table t;
    t.n = n;
    if (hipMalloc((void**)&t.d, 64 * n) != hipSuccess) return fail(c, H2AGG_ERR_NOMEM, "hipMalloc(base table)");
    int rc = ensure(c, c->tmp_bases, 64 * n);
    if (rc == H2AGG_OK) {
        hipError_t e = hipMemcpyAsync(c->tmp_bases.p, bases, 64 * n, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = fail(c, H2AGG_ERR_HIP, hipGetErrorString(e));
    }
    if (rc == H2AGG_OK) rc = clear_flags(c);
    if (rc == H2AGG_OK) {
        hipLaunchKernelGGL(k_bases_to_mont, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream,
                           (const uint8_t*)c->tmp_bases.p, n, t.d, c->d_flags);
        rc = finish(c);
    }
    if (rc != H2AGG_OK) {
        hipFree(t.d);
        return rc;
    }
    uint64_t h = c->next_handle++;
    c->tables[h] = t;
    *handle_out = h;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_bases_generate(h2agg_ctx* c, const void* d_k, size_t n, uint64_t* handle_out) try {
    TRY(bind(c));
    if (!d_k || !handle_out || n == 0) return fail(c, H2AGG_ERR_INVALID, "null buffer or n == 0");
    Table t;
    t.n = n;
    if (hipMalloc((void**)&t.d, 64 * n) != hipSuccess) return fail(c, H2AGG_ERR_NOMEM, "hipMalloc(base table)");
    int rc = clear_flags(c);
    if (rc == H2AGG_OK) {
        static const bool ladder = knob("H2AGG_SCALAR_MUL") && !strcmp(knob("H2AGG_SCALAR_MUL"), "ladder");
        size_t blocks = (n + BLOCK - 1) / BLOCK;
        if (blocks > 65535 * 16) blocks = 65535 * 16;
        if (ladder) {
            hipLaunchKernelGGL(k_bases_generate, dim3((unsigned)blocks), dim3(BLOCK), 0, c->stream, (const uint8_t*)d_k, n, t.d,
                               c->d_flags);
        } else {
            // fixed-base comb: d * 2^(8w) * G for every byte position, built once per context (32 x 255 points, 510 KiB)
            if (!c->comb_ready) {
                rc = ensure(c, c->comb, (size_t)COMB_WINDOWS * COMB_ROW * 64);
                if (rc == H2AGG_OK) {
                    hipLaunchKernelGGL(k_comb_table_build, dim3((COMB_WINDOWS * COMB_ROW + SM_GROUPS - 1) / SM_GROUPS),
                                       dim3(SM_THREADS), 0, c->stream, (uint8_t*)c->comb.p, c->d_flags);
                    c->comb_ready = true;
                }
            }
            if (rc == H2AGG_OK)
                hipLaunchKernelGGL(k_bases_generate_comb, dim3((unsigned)blocks), dim3(BLOCK), 0, c->stream, (const uint8_t*)d_k, n,
                                   (const uint8_t*)c->comb.p, t.d, c->d_flags);
        }
        if (rc == H2AGG_OK) rc = finish(c);
    }
    if (rc != H2AGG_OK) {
        hipFree(t.d);
        return rc;
    }
    uint64_t h = c->next_handle++;
    c->tables[h] = t;
    *handle_out = h;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_bases_download(h2agg_ctx* c, uint64_t handle, size_t first, size_t n, uint8_t* out) try {
    TRY(bind(c));
    auto it = c->tables.find(handle);
    if (it == c->tables.end()) return fail(c, H2AGG_ERR_INVALID, "unknown base-table handle");
    if (!out || first + n > it->second.n) return fail(c, H2AGG_ERR_INVALID, "range outside the table");
    if (n == 0) return H2AGG_OK;
    TRY(ensure(c, c->out, 64 * n));
    hipLaunchKernelGGL(k_bases_from_mont, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, it->second.d + 64 * first,
                       n, (uint8_t*)c->out.p);
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, 64 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_bases_free(h2agg_ctx* c, uint64_t handle) try {
    TRY(bind(c));
    auto it = c->tables.find(handle);
    if (it == c->tables.end()) return fail(c, H2AGG_ERR_INVALID, "unknown base-table handle");
    TRY(join_tails(c));   // tails and accumulations on the context's other streams (and a deferred tail) may still read the table
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipFree(it->second.d);
    if (it->second.endo_x) hipFree(it->second.endo_x);
    if (it->second.pre) hipFree(it->second.pre);
    it->second.free_combs();
    c->tables.erase(it);
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ---------------------------------------------------------------- MSM
}  // extern "C"
namespace {
// beta * x column of a resident table, made once
int table_endo(h2agg_ctx* c, Table& t, const uint8_t** out) {
    if (!t.endo_x) {
        if (hipMalloc((void**)&t.endo_x, 32 * t.n) != hipSuccess) return fail(c, H2AGG_ERR_NOMEM, "hipMalloc(endo table)");
        hipLaunchKernelGGL(k_bases_endo_x, dim3(grid_for(c, t.n)), dim3(BLOCK), 0, c->stream, (const uint8_t*)t.d, t.n,
                           t.endo_x);
    }
    *out = t.endo_x;
    return H2AGG_OK;
}
}  // namespace
extern "C" {
// width of the fixed-base levels: n * ceil(255 / c) bucket insertions + one reduction of 2^(c-1) buckets (~4 mixed-add
// equivalents per bucket), minimised over c; capped so the sort's packed items and partitions stay in range
int choose_pre_window(size_t n) {
    int best = 16;
    double best_cost = 1e300;
    for (int cc = 8; cc <= 20; ++cc) {
        const double cost = (double)n * ((255 + cc - 1) / cc) + 4.0 * (double)((size_t)1 << (cc - 1));
        if (cost < best_cost) {
            best_cost = cost;
            best = cc;
        }
    }
    return best;
}

int h2agg_bases_precompute(h2agg_ctx* c, uint64_t handle, int window_bits) try {
    TRY(bind(c));
    auto it = c->tables.find(handle);
    if (it == c->tables.end()) return fail(c, H2AGG_ERR_INVALID, "unknown base-table handle");
    Table& t = it->second;
    if (window_bits != 0 && (window_bits < 4 || window_bits > 20))
        return fail(c, H2AGG_ERR_INVALID, "fixed-base window must be 0 (auto) or in [4, 20]");
    int cc = window_bits ? window_bits : choose_pre_window(t.n);
    // tables whose levels outgrow the packed sort item take c = 20 and the (level, point) sort of fb_sort_kernels.hpp
    if (!window_bits && (size_t)((255 + cc - 1) / cc) * t.n > ((size_t)1 << 22)) cc = FB_C;
    const int W = (255 + cc - 1) / cc;
    if (cc == FB_C && t.n > (size_t)FB_MAX_TILES * FB_T)
        return fail(c, H2AGG_ERR_INVALID, "table too large for fixed-base levels (at most 2^22 points)");
    // other widths are addressed through the sort's packed 32-bit item (22 index bits next to 9 sub-bucket bits + sign):
    // beyond that the sort falls back to its two-array kernels and the levels stop paying for themselves (round 4, with the
    // limit lifted: 16 MSMs over a 2^22-point table 74 ms on the ordinary path, 113 ms through levels at c = 20 + the two-array
    // sort).  debug key "pre_big" lifts the limit for A/B runs of exactly that.
    if (cc != FB_C && (size_t)W * t.n > ((size_t)1 << 22) && !c->dbg_pre_big)
        return fail(c, H2AGG_ERR_INVALID, "table too large for fixed-base levels (ceil(255 / c) * n must be <= 2^22)");
    if (!window_bits) {
        // auto mode is the opportunistic call: it does not take a large share of what is left of the device for levels
        // (832 B per point at c = 20: 3.25 GiB for a 2^22-point table) — the caller who wants them regardless names the width
        size_t free_b = 0, total_b = 0;
        HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
        if ((size_t)W * t.n * 64 > free_b / 4)
            return fail(c, H2AGG_ERR_NOMEM, "fixed-base levels (auto) would take more than a quarter of the free device memory; pass window_bits to insist");
    }
    TRY(join_tails(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (t.pre) {
        hipFree(t.pre);
        t.pre = nullptr;
    }
    t.free_combs();
    if (hipMalloc((void**)&t.pre, (size_t)W * t.n * 64) != hipSuccess)
        return fail(c, H2AGG_ERR_NOMEM, "hipMalloc(fixed-base levels)");
    HIP_TRY(c, hipMemcpyAsync(t.pre, t.d, t.n * 64, hipMemcpyDeviceToDevice, c->stream));
    for (int w = 1; w < W; ++w)
        hipLaunchKernelGGL(k_bases_shift, dim3(grid_for(c, t.n)), dim3(BLOCK), 0, c->stream,
                           (const uint8_t*)t.pre + (size_t)(w - 1) * t.n * 64, t.n, cc, t.pre + (size_t)w * t.n * 64);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    t.pre_c = cc;
    t.pre_W = W;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

}   // extern "C"
namespace {
// `batch` MSMs of n <= COMB_MSM_MAX scalars each over the leading bases of a table with fixed-base levels: through the table's
// comb (made here on first use).  Returns false when the comb route does not apply.
bool comb_msm_applies(const h2agg_ctx* c, const Table& t, size_t n) {
    const bool off = !c->dbg_comb_msm;
    return !off && t.pre && !c->cfg_c && n >= 1 && n <= (size_t)COMB_MSM_MAX;
}
// *done = false: the comb could not be made (no memory for it): the caller takes the bucket path, which needs none.
int comb_msm_run(h2agg_ctx* c, Table& t, const uint8_t* d_scalars, size_t n, size_t batch, uint8_t* d_out_jac, bool* done) {
    *done = true;
    if (!t.comb || t.comb_n < n) {
        // sized for what is asked (a one-point multi_exp does not pay for 256 bases' rows: 520 KiB per base), doubling so that
        // a caller whose sizes creep up rebuilds O(log) times; everything is ordered by the context's stream, no host wait:
        // the comb it replaces may still be read by kernels in flight and is only retired (Table::free_combs)
        const size_t cap_n = t.n < (size_t)COMB_MSM_MAX ? t.n : (size_t)COMB_MSM_MAX;
        size_t nb = 16;
        while (nb < n) nb *= 2;
        if (nb > cap_n) nb = cap_n;
        const size_t entries = nb * (size_t)COMB_WINDOWS * COMB_ROW;
        uint8_t* fresh = nullptr;
        if (hipMalloc((void**)&fresh, entries * 64) != hipSuccess) {
            (void)hipGetLastError();
            *done = false;
            return H2AGG_OK;
        }
        if (t.comb) t.comb_retired.push_back(t.comb);
        t.comb = fresh;
        size_t grid = (entries + SM_GROUPS - 1) / SM_GROUPS;
        const size_t cap = (size_t)c->cu_count * 16;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(k_comb_table_build, dim3((unsigned)grid), dim3(SM_THREADS), 0, c->stream, t.comb, c->d_flags,
                           (const uint8_t*)t.d, (uint32_t)nb);
        t.comb_n = nb;
    }
    hipLaunchKernelGGL(k_comb_msm, dim3((unsigned)batch), dim3(BLOCK), 0, c->stream, (const uint8_t*)t.comb, d_scalars, (uint32_t)n,
                       d_out_jac, c->d_flags);
    HIP_TRY(c, hipGetLastError());
    return H2AGG_OK;
}
}   // namespace
extern "C" {

int h2agg_g1_msm_device_async(h2agg_ctx* c, uint64_t handle, const void* d_scalars, size_t n, void* d_out_jac) try {
    TRY(bind(c));
    auto it = c->tables.find(handle);
    if (it == c->tables.end()) return fail(c, H2AGG_ERR_INVALID, "unknown base-table handle");
    if (!d_scalars || !d_out_jac) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    if (n == 0) return fail(c, H2AGG_ERR_EMPTY, "multi_exp of zero pairs (reference panics: mock/arith/ecc.rs:128)");
    if (n > it->second.n) return fail(c, H2AGG_ERR_INVALID, "more scalars than bases in the table");
    if (comb_msm_applies(c, it->second, n)) {
        bool done = false;
        TRY(comb_msm_run(c, it->second, (const uint8_t*)d_scalars, n, 1, (uint8_t*)d_out_jac, &done));
        if (done) return H2AGG_OK;
    }
    const uint8_t* endo = nullptr;
    TRY(table_endo(c, it->second, &endo));
    // Beyond 2^22 points the packed (index | sub-bucket) sort item no longer fits 32 bits and the sort would fall back to
    // its slower two-array kernels: cut the MSM into slices of <= 2^22 points instead (their tails overlap the next
    // slice's bulk) and add the slices' results.
    if (it->second.pre && !c->cfg_c) {   // fixed-base levels (h2agg_bases_precompute); an explicit window_bits overrides
        const PreTable pt{it->second.pre, it->second.n, it->second.pre_c, it->second.pre_W};
        if ((size_t)pt.W * n < ((size_t)1 << 32))
            return msm_run(c, it->second.d, (const uint8_t*)d_scalars, n, (uint8_t*)d_out_jac, 1, nullptr, &pt);
    }
    const size_t SLICE = (size_t)1 << 22;
    if (n <= SLICE) return msm_run(c, it->second.d, (const uint8_t*)d_scalars, n, (uint8_t*)d_out_jac, 1, endo);
    // The slices share ONE bucket set (chain modes of msm_run): every slice adds its points to the sums the buckets already
    // hold, and only the last one is followed by the bucket reduction / window sums / Horner tail.
    const size_t nsl = (n + SLICE - 1) / SLICE;
    const bool was_overlap = c->tail_overlap;
    const int was_level = c->overlap_level;
    c->tail_overlap = true;
    c->overlap_level = 2;
    c->chain_n = n;
    c->chain_glv = false;
    int rc = H2AGG_OK;
    for (size_t k = 0; k < nsl && rc == H2AGG_OK; ++k) {
        const size_t off = k * SLICE, m = n - off < SLICE ? n - off : SLICE;
        c->chain = k == 0 ? CHAIN_FIRST : (k + 1 == nsl ? CHAIN_LAST : CHAIN_MID);
        rc = msm_run(c, it->second.d + 64 * off, (const uint8_t*)d_scalars + 32 * off, m, (uint8_t*)d_out_jac, 1, endo + 32 * off);
    }
    c->chain = CHAIN_OFF;
    c->tail_overlap = was_overlap;
    c->overlap_level = was_level;
    TRY(rc);
    if (!was_overlap) TRY(join_tails(c));   // without overlap mode the caller's stream orders the result
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_msm_device_batch_async(h2agg_ctx* c, uint64_t handle, const void* d_scalars, size_t n, size_t batch,
                                    void* d_out_jac) try {
    TRY(bind(c));
    auto it = c->tables.find(handle);
    if (it == c->tables.end()) return fail(c, H2AGG_ERR_INVALID, "unknown base-table handle");
    if (!d_scalars || !d_out_jac) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    if (n == 0 || batch == 0)
        return fail(c, H2AGG_ERR_EMPTY, "multi_exp of zero pairs (reference panics: mock/arith/ecc.rs:128)");
    if (n > it->second.n) return fail(c, H2AGG_ERR_INVALID, "more scalars than bases in the table");
    if (comb_msm_applies(c, it->second, n)) {
        bool done = false;
        TRY(comb_msm_run(c, it->second, (const uint8_t*)d_scalars, n, batch, (uint8_t*)d_out_jac, &done));
        if (done) return H2AGG_OK;
    }
    // MSMs per set of launches: the level-1 sort partitions (batch * W * NB >> sub_bits, sub_bits <= 11) must fit its
    // LDS counters, and the entry count its 32-bit offsets
    // fixed-base levels (h2agg_bases_precompute).  Levels at c = 20 pay only through their own sort, which takes one MSM at
    // a time and no sort knobs: where it does not apply (a configured context; a batch of short columns over a big table, which
    // would also want batch x 2^19 buckets) the levels are left alone and the ordinary path runs over the table — the two-array
    // sort over c = 20 levels measured 113 ms against the ordinary path's 74 (16 MSMs, 2^22 points).
    const bool fb_table = it->second.pre && it->second.pre_c == FB_C && !c->dbg_pre_big;
    const bool use_pre = it->second.pre && !c->cfg_c && (!fb_table || (fb_sort_knobs_clear(c) && (batch == 1 || n >= ((size_t)1 << 18))));
    const PreTable pt{it->second.pre, it->second.n, it->second.pre_c, it->second.pre_W};
    const MsmPlan p1 = make_plan(c, n, 1);
    const uint32_t nb1 = use_pre ? (1u << (pt.c - 1)) : p1.NB;
    const size_t sets1 = use_pre ? 1 : (size_t)p1.W;                               // bucket sets per MSM
    uint32_t ppw_min = nb1 >> 11;
    if (ppw_min < 1) ppw_min = 1;
    size_t per = (size_t)SORT_MAX_PW / (sets1 * ppw_min);
    const size_t ent1 = use_pre ? n * (size_t)pt.W : n * (size_t)p1.W * (p1.glv ? 2 : 1);
    if (per * ent1 >= ((size_t)1 << 31)) per = (((size_t)1 << 31) - 1) / ent1;
    if (per * n >= ((size_t)1 << 29)) per = (((size_t)1 << 29) - 1) / n;
    if (per < 1) per = 1;
    // From 2^20 points on an MSM fills the chip by itself and takes the digit-major sort / two-dimensional reduction, which
    // batches do not: one MSM after another, each one's tail under the next one's bulk (2^22 points: 6.1 -> 5.4 ms each)
    const bool one_by_one = use_pre ? (pt.c == FB_C && n >= ((size_t)1 << 18)) : n >= ((size_t)1 << 20);
    if (one_by_one) per = 1;
    const uint8_t* endo = nullptr;
    TRY(table_endo(c, it->second, &endo));
    const bool was_overlap = c->tail_overlap;
    const int was_level = c->overlap_level;
    if (one_by_one && batch > 1) {
        c->tail_overlap = true;
        c->overlap_level = 2;
    }
    int rc = H2AGG_OK;
    for (size_t q = 0; q < batch && rc == H2AGG_OK; q += per) {
        const size_t b = batch - q < per ? batch - q : per;
        rc = msm_run(c, it->second.d, (const uint8_t*)d_scalars + 32 * n * q, n, (uint8_t*)d_out_jac + 96 * q, (uint32_t)b,
                     use_pre ? nullptr : endo, use_pre ? &pt : nullptr);
    }
    c->tail_overlap = was_overlap;
    c->overlap_level = was_level;
    TRY(rc);
    if (one_by_one && batch > 1 && !was_overlap) TRY(join_tails(c));   // without overlap mode the caller's stream orders the results
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_msm_device(h2agg_ctx* c, uint64_t handle, const void* d_scalars, size_t n, uint8_t out[96]) try {
    TRY(bind(c));
    if (!out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    set_identity_jac(out);
    TRY(clear_flags(c));
    TRY(h2agg_g1_msm_device_async(c, handle, d_scalars, n, c->d_res_jac));
    return fetch_result_jac(c, out);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_g1_msm_preloaded(h2agg_ctx* c, uint64_t handle, const uint8_t* scalars, size_t n, uint8_t out[96]) try {
    TRY(bind(c));
    if (!out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    set_identity_jac(out);
    if (n == 0) return fail(c, H2AGG_ERR_EMPTY, "multi_exp of zero pairs (reference panics: mock/arith/ecc.rs:128)");
    if (!scalars) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_b, 32 * n));
    HIP_TRY(c, hipMemcpyAsync(c->in_b.p, scalars, 32 * n, hipMemcpyHostToDevice, c->stream));
    return h2agg_g1_msm_device(c, handle, c->in_b.p, n, out);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// assign_instance_commitment for one instance column (verify.rs:601-603, 623-639)
int h2agg_instance_commitment(h2agg_ctx* c, uint64_t g_lagrange_handle, const uint8_t* instance, size_t len,
                              size_t max_len, uint8_t out_jac[96]) try {
    TRY(bind(c));
    if (!out_jac) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    set_identity_jac(out_jac);
    if (len > max_len)
        return fail(c, H2AGG_ERR_INVALID, "assert!(instance.len() <= params.n() - (blinding_factors + 1)) failed (verify.rs:601-603)");
    if (len == 0) return H2AGG_OK;  // None => pchip.assign_const(identity)  (verify.rs:638)
    return h2agg_g1_msm_preloaded(c, g_lagrange_handle, instance, len, out_jac);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_host_alloc(h2agg_ctx* c, size_t bytes, void** out) try {
    TRY(bind(c));
    if (!out || bytes == 0) return fail(c, H2AGG_ERR_INVALID, "null out or zero size");
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, H2AGG_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}
int h2agg_host_free(h2agg_ctx* c, void* p) try {
    TRY(bind(c));
    if (p) HIP_TRY(c, hipHostFree(p));
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

}   // extern "C"
namespace {
// h2agg_g1_msm / h2agg_g1_msm_jac: host buffers in.  stride = 64: canonical affine bases; 96: canonical Jacobian points,
// normalised on the device (k_jac_to_mont_affine) where the affine entry point converts to Montgomery form (k_bases_to_mont)
int msm_host(h2agg_ctx* c, const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, uint8_t out[96]);
}
extern "C" {
int h2agg_g1_msm(h2agg_ctx* c, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[96]) try {
    return msm_host(c, bases, 64, scalars, n, out);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}
int h2agg_g1_msm_jac(h2agg_ctx* c, const uint8_t* points_jac, const uint8_t* scalars, size_t n, uint8_t out[96]) try {
    return msm_host(c, points_jac, 96, scalars, n, out);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;
} catch (...) {
    return H2AGG_ERR_INVALID;
}
}   // extern "C"
namespace {
int msm_host(h2agg_ctx* c, const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, uint8_t out[96]) {
    TRY(bind(c));
    if (!out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    set_identity_jac(out);
    if (n == 0) return fail(c, H2AGG_ERR_EMPTY, "multi_exp of zero pairs (reference panics: mock/arith/ecc.rs:128)");
    if (!bases || !scalars) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    TRY(ensure(c, c->in_a, stride * n));
    TRY(ensure(c, c->in_b, 32 * n));
    TRY(ensure(c, c->tmp_bases, 64 * n));
    TRY(clear_flags(c));
    // Large inputs are cut into slices that cross PCIe on a copy stream while the previous slice is being computed
    // (an MSM is a sum over points, so the slices' results just add up): 96 B/point of transfer hide under ~1.7 ns/point
    // of arithmetic, instead of preceding it.  Slices of >= 2^19 points keep the per-MSM efficiency (2^18-point slices lose more than the overlap gains).
    const bool chain_env_off = !c->dbg_pcie_chain;
    const bool chained = !chain_env_off;
    // chained slices (below) of ~200 K points: the transfer (1.73 ns/point) and the slice's sort + accumulation (~90 us +
    // 1.3 ns/point) then take turns of equal length — measured, profiles/r03_sweeps.txt section 10: 2^20 points 3.03 / 2.97 /
    // 2.92 / 2.96 / 3.15 ms with 3 / 4 / 5 / 6 / 8 slices
    const size_t MIN_SLICE = chained ? (size_t)200 << 10 : (size_t)1 << 19;
    size_t nslices = chained ? (n + MIN_SLICE / 2) / MIN_SLICE : n / MIN_SLICE;
    if (nslices > MSM_MAX_SLICES) nslices = MSM_MAX_SLICES;
    const int env_slices = c->dbg_pcie_slices;   // (h2agg_debug_configure "pcie_slices": tests vary it per call)
    if (env_slices >= 1 && env_slices <= MSM_MAX_SLICES && n >= ((size_t)1 << 10) * (size_t)env_slices) nslices = (size_t)env_slices;
    if (nslices < 2) {
        HIP_TRY(c, hipMemcpyAsync(c->in_a.p, bases, stride * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->in_b.p, scalars, 32 * n, hipMemcpyHostToDevice, c->stream));
        if (stride == 96)
            hipLaunchKernelGGL(k_jac_to_mont_affine, dim3(grid_for(c, (n + TA_K - 1) / TA_K)), dim3(BLOCK), 0, c->stream,
                               (const uint8_t*)c->in_a.p, n, (uint8_t*)c->tmp_bases.p, c->d_flags);
        else
            hipLaunchKernelGGL(k_bases_to_mont, dim3(grid_for(c, n)), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, n,
                               (uint8_t*)c->tmp_bases.p, c->d_flags);
        TRY(msm_run(c, (const uint8_t*)c->tmp_bases.p, (const uint8_t*)c->in_b.p, n, c->d_res_jac));
        return fetch_result_jac(c, out);
    }
    TRY(ensure(c, c->out, 96 * MSM_MAX_SLICES));
    // the copy stream must not overwrite in_a / in_b while earlier work on the main stream still reads them
    HIP_TRY(c, hipEventRecord(c->ev_ready, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_ready, 0));
    const size_t per = (n + nslices - 1) / nslices;
    const bool was_overlap = c->tail_overlap;
    const int was_level = c->overlap_level;
    c->tail_overlap = true;
    c->overlap_level = 2;
    // The slices share one bucket set (chain modes of msm_run): slice k only sorts its keys and adds its points to the bucket
    // sums; ONE reduction / window-sum / Horner chain follows the last slice — that chain (0.6-0.9 ms of pure latency) is what
    // is left exposed behind the last byte of the transfer, so the slices can be small.  H2AGG_PCIE_CHAIN=0: the earlier
    // scheme (every slice a whole MSM, results added) for A/B runs.
    const int chain_glv_env = c->dbg_pcie_glv;
    if (chained) {
        c->chain_n = n;
        c->chain_glv = chain_glv_env ? chain_glv_env > 0 : (c->cfg_glv >= 0 && n < ((size_t)1 << 22));
    }
    int rc = H2AGG_OK;
    size_t done = 0, k = 0;
    for (; done < n && rc == H2AGG_OK; done += per, ++k) {
        const size_t m = n - done < per ? n - done : per;
        chaos_wait(16, c->copy_stream);
        hipError_t e = hipMemcpyAsync((uint8_t*)c->in_b.p + 32 * done, scalars + 32 * done, 32 * m, hipMemcpyHostToDevice,
                                      c->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev_copy_s[k], c->copy_stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync((uint8_t*)c->in_a.p + stride * done, bases + stride * done, stride * m, hipMemcpyHostToDevice,
                               c->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev_copy[k], c->copy_stream);
        // the sort of a slice only reads its scalars: it runs while the slice's bases are still on their way (chained
        // slices; the earlier scheme waits for both, as it always did)
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, chained ? c->ev_copy_s[k] : c->ev_copy[k], 0);
        if (e != hipSuccess) {
            rc = fail(c, H2AGG_ERR_HIP, hipGetErrorString(e));
            break;
        }
        auto convert = [=](hipStream_t st) {
            if (stride == 96)
                hipLaunchKernelGGL(k_jac_to_mont_affine, dim3(grid_for(c, (m + TA_K - 1) / TA_K)), dim3(BLOCK), 0, st,
                                   (const uint8_t*)c->in_a.p + 96 * done, m, (uint8_t*)c->tmp_bases.p + 64 * done, c->d_flags);
            else
                hipLaunchKernelGGL(k_bases_to_mont, dim3(grid_for(c, m)), dim3(BLOCK), 0, st,
                                   (const uint8_t*)c->in_a.p + 64 * done, m, (uint8_t*)c->tmp_bases.p + 64 * done, c->d_flags);
        };
        if (chained) {
            c->chain = done == 0 ? CHAIN_FIRST : (done + per >= n ? CHAIN_LAST : CHAIN_MID);
            hipEvent_t ev_bases = c->ev_copy[k];
            c->bases_hook = [=](hipStream_t st) -> int {
                HIP_TRY(c, hipStreamWaitEvent(st, ev_bases, 0));
                convert(st);
                return H2AGG_OK;
            };
        } else {
            convert(c->stream);
        }
        rc = msm_run(c, (const uint8_t*)c->tmp_bases.p + 64 * done, (const uint8_t*)c->in_b.p + 32 * done, m,
                     chained ? c->d_res_jac : (uint8_t*)c->out.p + 96 * k);
        c->bases_hook = nullptr;
    }
    c->chain = CHAIN_OFF;
    c->tail_overlap = was_overlap;
    c->overlap_level = was_level;
    TRY(rc);
    TRY(join_tails(c));
    if (!chained)
        hipLaunchKernelGGL(k_g1_sum, dim3(1), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->out.p, k, c->d_res_jac, c->d_flags);
    return fetch_result_jac(c, out);
}
}   // namespace
extern "C" {

int h2agg_eval_flat(h2agg_ctx* c, const uint8_t* pts, const uint8_t* scalars, const uint8_t* has_scalar, size_t n,
                    uint8_t out[96]) try {
    TRY(bind(c));
    if (!out) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    set_identity_jac(out);
    if (n && (!pts || !scalars || !has_scalar)) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    // host-side partition of the flat list (evaluation.rs:189-196): entries with / without a scalar
    std::vector<uint8_t> ps, ss, pn;
    for (size_t i = 0; i < n; ++i) {
        if (has_scalar[i]) {
            ps.insert(ps.end(), pts + 64 * i, pts + 64 * i + 64);
            ss.insert(ss.end(), scalars + 32 * i, scalars + 32 * i + 32);
        } else {
            pn.insert(pn.end(), pts + 64 * i, pts + 64 * i + 64);
        }
    }
    const size_t m = ss.size() / 32, k = pn.size() / 64;
    if (m == 0) return fail(c, H2AGG_ERR_EMPTY, "multi_exp of zero pairs (reference panics: mock/arith/ecc.rs:128)");
    TRY(ensure(c, c->in_a, 64 * m));
    TRY(ensure(c, c->in_b, 32 * m));
    TRY(ensure(c, c->in_c, 64 * k + 64));
    TRY(ensure(c, c->tmp_bases, 64 * m));
    HIP_TRY(c, hipMemcpyAsync(c->in_a.p, ps.data(), 64 * m, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_b.p, ss.data(), 32 * m, hipMemcpyHostToDevice, c->stream));
    if (k) HIP_TRY(c, hipMemcpyAsync(c->in_c.p, pn.data(), 64 * k, hipMemcpyHostToDevice, c->stream));
    // pageable-host sources must stay alive until the copies are done
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    TRY(clear_flags(c));
    hipLaunchKernelGGL(k_bases_to_mont, dim3(grid_for(c, m)), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->in_a.p, m,
                       (uint8_t*)c->tmp_bases.p, c->d_flags);
    TRY(msm_run(c, (const uint8_t*)c->tmp_bases.p, (const uint8_t*)c->in_b.p, m, nullptr));
    TRY(join_tails(c));
    hipLaunchKernelGGL(k_eval_tail, dim3(1), dim3(BLOCK), 0, c->stream, (const uint8_t*)c->d_res_xyzz,
                       (const uint8_t*)c->in_c.p, k, c->d_res_jac, c->d_flags);
    return fetch_result_jac(c, out);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ---------------------------------------------------------------- tuning / measurement
int h2agg_msm_configure(h2agg_ctx* c, int window_bits, int reduce_segment, int big_bucket_threshold) try {
    if (!c) return H2AGG_ERR_INVALID;
    if (window_bits != 0 && (window_bits < 2 || window_bits > 20))
        return fail(c, H2AGG_ERR_INVALID, "window_bits must be 0 or in [2, 20]");
    if (reduce_segment < 0 || (reduce_segment & (reduce_segment - 1)))
        return fail(c, H2AGG_ERR_INVALID, "reduce_segment must be 0 or a power of two");
    if (big_bucket_threshold < 0) return fail(c, H2AGG_ERR_INVALID, "big_bucket_threshold must be >= 0");
    c->cfg_c = window_bits;
    c->cfg_seg = reduce_segment;
    c->cfg_big = big_bucket_threshold;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_msm_configure_glv(h2agg_ctx* c, int mode) try {
    if (!c) return H2AGG_ERR_INVALID;
    if (mode < -1 || mode > 1) return fail(c, H2AGG_ERR_INVALID, "mode must be -1 (off), 0 (auto) or 1 (on)");
    c->cfg_glv = mode;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}
int h2agg_msm_configure_lanes_per_bucket(h2agg_ctx* c, int lanes) try {
    if (!c) return H2AGG_ERR_INVALID;
    if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 4 && lanes != 8 && lanes != 16)
        return fail(c, H2AGG_ERR_INVALID, "lanes must be 0, 1, 2, 4, 8 or 16");
    c->cfg_lpb = lanes;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_msm_configure_sort(h2agg_ctx* c, int sub_bits, int tile) try {
    if (!c) return H2AGG_ERR_INVALID;
    if (sub_bits != 0 && (sub_bits < 4 || sub_bits > SORT_MAX_SUB_BITS))
        return fail(c, H2AGG_ERR_INVALID, "sub_bits must be 0 or in [4, 12]");
    if (tile != 0 && tile != -1 && tile != -2 && tile != -3 && (tile < BLOCK || tile > (1 << 16)))
        return fail(c, H2AGG_ERR_INVALID, "tile must be 0, -1, -2 or in [256, 65536]");
    c->cfg_sub_bits = sub_bits;
    c->cfg_no_stage = tile == -1;   // -1: force the direct two-array sort kernels
    c->cfg_no_dm = tile == -3;      // -3: packed two-level sort even where the digit-major one applies
    c->cfg_stage_l1 = tile == -2;   // -2: also stage level 1 through LDS (experiment: slower, kept for tests)
    c->cfg_tile = tile > 0 ? tile : 0;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_debug_configure(h2agg_ctx* c, const char* key, int value) try {
    if (!c || !key) return H2AGG_ERR_INVALID;
    const std::string k(key);
    if (k == "pcie_slices") c->dbg_pcie_slices = value;
    else if (k == "pcie_glv") c->dbg_pcie_glv = value;
    else if (k == "pcie_chain") c->dbg_pcie_chain = value;
    else if (k == "comb_msm") c->dbg_comb_msm = value;
    else if (k == "plan_cache") c->dbg_plan_cache = value;
    else if (k == "small_sort") c->dbg_small_sort = value;   // 0: small MSMs take the packed two-level sort again
    else if (k == "eval_split") c->dbg_eval_split = value;   // 0: an evaluation's two multi_exps are two MSMs again
    else if (k == "shard_fail") c->dbg_shard_fail = value;   // tests: this rank of a sharded aggregation fails before (1) / between (2) the exchanges, or inside exchange 1 (3) / 2 (4) before its all-gather
    else if (k == "lean_acc") {   // 0: the bucket accumulation through the generic kernel (k_msm_accumulate) instead of the lean one
#ifdef H2AGG_MEASURE_KNOBS
        c->dbg_lean_acc = value;
#else
        if (value != 1) return fail(c, H2AGG_ERR_INVALID, "h2agg_debug_configure: the generic accumulation kernel is in the measure build only (build_ext.py --measure)");
#endif
    }
    else if (k == "tape_lds") {
        if (c->dbg_tape_lds != value) agg_plans_release(c);   // (kept recordings carry their tape in one encoding or the other)
        c->dbg_tape_lds = value;
    }       // 0: every Fr tape through k_tape_run (register file in L2) instead of the LDS one
    else if (k == "prewake") c->dbg_prewake = value;         // 0: the sponge workers sleep until their chains are posted (A/B of the pre-wake)
    else if (k == "phases") c->dbg_phases = value;           // 1: every h2agg_verify_aggregation* call keeps its wall-clock split for h2agg_last_phases
    else if (k == "pre_big") c->dbg_pre_big = value;         // 1: h2agg_bases_precompute takes any explicit width (levels through the two-array sort)
    else return fail(c, H2AGG_ERR_INVALID, "h2agg_debug_configure: unknown key " + k);
    return H2AGG_OK;
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_msm_set_tail_overlap(h2agg_ctx* c, int enable) try {
    TRY(bind(c));
    TRY(join_tails(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tail_overlap = enable != 0;
    if (enable >= 1 && enable <= 3) c->overlap_level = enable;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_profile_enable(h2agg_ctx* c, int enable) try {
    TRY(bind(c));
    if (enable && !c->prof_events_created) {
        for (int r = 0; r < h2agg_ctx::PROF_RING; ++r)
            for (int s = 0; s < ST_N; ++s)
                for (int k = 0; k < 2; ++k) HIP_TRY(c, hipEventCreateWithFlags(&c->prof[r].ev[s][k], EV_TIME_FLAGS));
        c->prof_events_created = true;
    }
    if (!enable) profile_harvest_all(c);
    c->profiling = enable != 0;
    c->prof_only = (enable >= 2 && enable - 2 < ST_N) ? enable - 2 : -1;
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}
int h2agg_profile_reset(h2agg_ctx* c) try {
    if (!c) return H2AGG_ERR_INVALID;
    profile_harvest_all(c);
    for (int s = 0; s < ST_N; ++s) {
        c->stage_ms[s] = 0;
        c->stage_launches[s] = 0;
    }
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}
int h2agg_profile_stage_count(h2agg_ctx*) { return ST_N; }
const char* h2agg_profile_stage_name(h2agg_ctx*, int i) { return (i >= 0 && i < ST_N) ? STAGE_NAMES[i] : ""; }
int h2agg_profile_stage_get(h2agg_ctx* c, int i, double* total_ms, uint64_t* launches) try {
    if (!c || i < 0 || i >= ST_N) return H2AGG_ERR_INVALID;
    profile_harvest_all(c);
    if (total_ms) *total_ms = c->stage_ms[i];
    if (launches) *launches = c->stage_launches[i];
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ---------------------------------------------------------------- pairing check (host)
extern "C++" {
namespace {
// csrc/transcript.inc (HostPool): if a pool worker is spinning for work, `on_worker` runs there while `here` runs on the
// calling thread, both done on return (true); otherwise nothing has run (false)
bool host_pool_pair_if_awake(const std::function<void()>& on_worker, const std::function<void()>& here);
bool pairing_adx_ok() {
#ifdef H2AGG_PAIRING_ADX_BUILD
    static const bool ok = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx") && !getenv("H2AGG_PAIRING_PORTABLE");
    return ok;
#else
    return false;
#endif
}
template <class P>
int pairing_load(h2agg_ctx* c, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, std::vector<typename P::G1>& ps,
                 std::vector<typename P::G2>& qs) {
    if (n && (!g1_aff || !g2_aff)) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    ps.resize(n);
    qs.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const int r1 = P::load1(g1_aff + 64 * i, ps[i]);
        if (r1 == 1) return fail(c, H2AGG_ERR_NONCANONICAL, "pairing: G1 coordinate >= p");
        if (r1) return fail(c, H2AGG_ERR_BAD_POINT, "pairing: G1 point not on the curve");
        const int r2 = P::load2(g2_aff + 128 * i, qs[i]);
        if (r2 == 1) return fail(c, H2AGG_ERR_NONCANONICAL, "pairing: G2 coordinate >= p");
        if (r2 == 2) return fail(c, H2AGG_ERR_BAD_POINT, "pairing: G2 point not on the twist");
        if (r2) return fail(c, H2AGG_ERR_BAD_POINT, "pairing: G2 point outside the order-r subgroup");
    }
    return H2AGG_OK;
}
template <class P>
int pairing_product_t(h2agg_ctx* c, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, uint8_t out_gt[384]) {
    std::vector<typename P::G1> ps;
    std::vector<typename P::G2> qs;
    TRY(pairing_load<P>(c, g1_aff, g2_aff, n, ps, qs));
    // (prepared lines: a G2 point that comes back — [s]_2, [1]_2 of one ParamsKZG — pays its doubling / addition steps once)
    std::vector<std::shared_ptr<const typename P::Prepared>> keep(n);
    std::vector<const typename P::Prepared*> preps(n);
    for (size_t i = 0; i < n; ++i) {
        keep[i] = P::prepared(g2_aff + 128 * i, false, qs[i]);
        preps[i] = keep[i].get();
    }
    P::product_prepared(ps, preps, out_gt);
    return H2AGG_OK;
}
template <class P>
int pairing_check_t(h2agg_ctx* c, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, int* ok) {
    std::vector<typename P::G1> ps;
    std::vector<typename P::G2> qs;
    TRY(pairing_load<P>(c, g1_aff, g2_aff, n, ps, qs));
    *ok = P::check(ps, qs) ? 1 : 0;
    return H2AGG_OK;
}
// e(left, [s]_2) * e(right, -[1]_2) == 1 ?   (verify.rs:733-739: `n_g2_prepared = -params.g2()`)
template <class P>
int final_pair_check_t(h2agg_ctx* c, const uint8_t left_aff[64], const uint8_t right_aff[64], const uint8_t s_g2[128], const uint8_t g2[128],
                       int* ok) {
    uint8_t g1s[128], g2s[256];
    memcpy(g1s, left_aff, 64);
    memcpy(g1s + 64, right_aff, 64);
    memcpy(g2s, s_g2, 128);
    memcpy(g2s + 128, g2, 128);
    std::vector<typename P::G1> ps;
    std::vector<typename P::G2> qs;
    TRY(pairing_load<P>(c, g1s, g2s, 2, ps, qs));
    P::negate(qs[1]);
    const std::shared_ptr<const typename P::Prepared> keep[2] = {P::prepared(s_g2, false, qs[0]), P::prepared(g2, true, qs[1])};
    const std::vector<const typename P::Prepared*> preps = {keep[0].get(), keep[1].get()};
    // With a pool worker already awake and waiting (h2agg_verify_aggregation wakes one while the evaluation is on the device),
    // the second pair's Miller loop runs there beside the first one's here: 64 squarings + 89 sparse products per thread
    // instead of 64 + 178 on one — the loop part of the check 0.25 -> 0.16 ms.  Without one (a busy pool: many calls in
    // flight; a direct call of this entry point) both pairs share one loop, which is less work in total.
    typename P::Gt f0, f1;
    const bool split = host_pool_pair_if_awake(
        [&] { f1 = P::miller_prepared(std::vector<typename P::G1>{ps[1]}, std::vector<const typename P::Prepared*>{preps[1]}); },
        [&] { f0 = P::miller_prepared(std::vector<typename P::G1>{ps[0]}, std::vector<const typename P::Prepared*>{preps[0]}); });
    *ok = (split ? P::check_product(f0, f1) : P::check_prepared(ps, preps)) ? 1 : 0;
    return H2AGG_OK;
}
}  // namespace
}  // extern "C++"

int h2agg_pairing_product(h2agg_ctx* c, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, uint8_t out_gt[384]) try {
    if (!out_gt) return fail(c, H2AGG_ERR_INVALID, "null buffer");
#ifdef H2AGG_PAIRING_ADX_BUILD
    if (pairing_adx_ok()) return pairing_product_t<pairing_adx::Api>(c, g1_aff, g2_aff, n, out_gt);
#endif
    return pairing_product_t<pairing::Api>(c, g1_aff, g2_aff, n, out_gt);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_pairing_check(h2agg_ctx* c, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, int* ok) try {
    if (!ok) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    *ok = 0;
#ifdef H2AGG_PAIRING_ADX_BUILD
    if (pairing_adx_ok()) return pairing_check_t<pairing_adx::Api>(c, g1_aff, g2_aff, n, ok);
#endif
    return pairing_check_t<pairing::Api>(c, g1_aff, g2_aff, n, ok);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

// ParamsKZG's g2 / s_g2 as stored by ParamsKZG::write (fs.rs:40-55 reads them back through halo2_proofs): 64-byte
// compressed G2 points -> the 128-byte affine form the pairing entry points take
int h2agg_g2_batch_decompress(h2agg_ctx* c, const uint8_t* in, size_t n, uint8_t* out_aff) try {
    if (n && (!in || !out_aff)) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    for (size_t i = 0; i < n; ++i) {
        pairing::G2Affine q;
        const int r = pairing::g2_decompress(in + 64 * i, q);
        if (r == 1) return fail(c, H2AGG_ERR_NONCANONICAL, "G2 coordinate >= p");
        if (r) return fail(c, H2AGG_ERR_BAD_POINT, "invalid G2 point encoding (x^3 + b is not a square)");
        pairing::g2_to_bytes(q, out_aff + 128 * i);
    }
    return H2AGG_OK;
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

int h2agg_final_pair_check(h2agg_ctx* c, const uint8_t left_aff[64], const uint8_t right_aff[64], const uint8_t s_g2[128],
                           const uint8_t g2[128], int* ok) try {
    if (!ok || !left_aff || !right_aff || !s_g2 || !g2) return fail(c, H2AGG_ERR_INVALID, "null buffer");
    *ok = 0;
#ifdef H2AGG_PAIRING_ADX_BUILD
    if (pairing_adx_ok()) return final_pair_check_t<pairing_adx::Api>(c, left_aff, right_aff, s_g2, g2, ok);
#endif
    return final_pair_check_t<pairing::Api>(c, left_aff, right_aff, s_g2, g2, ok);
} catch (const std::bad_alloc&) {
    return H2AGG_ERR_NOMEM;   // no C++ exception crosses the C ABI
} catch (...) {
    return H2AGG_ERR_INVALID;
}

}  // extern "C"

#include "schema_api.inc"
#include "transcript.inc"
#include "comm.inc"
#include "verifier.inc"
