// EvaluationQuerySchema on the GPU backend: host-side AST + symbolic eval_prepare, device-side Fr tape.
//
// Mirrors halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:
//   enum EvaluationQuerySchema { Commitment, Eval, Scalar, Add, Mul }   :15-27   (+ has_commitment :30-38)
//   impl Add / impl Mul (cache the children's has-commitment flags)      :62-84
//   eval                                                                 :172-203
//   eval_prepare                                                         :205-293
//   estimate                                                             :295-330
// and the tail of evaluate_multiopen_proof (verify.rs:705-731).
//
// The reference walks the tree calling schip.mul / schip.add (MockFieldChip = halo2curves Fr on one CPU
// thread).  Here the host walks the same tree with the same control flow, but every schip call only
// RECORDS an operation on a tape of Fr registers; the tape then runs on the device level by level (all
// operations whose inputs are ready run in parallel), the resulting scalars are gathered straight into
// the MSM's scalar buffer, and no field arithmetic happens on the host.  Key merging uses a hash index
// instead of the reference's linear `find` (evaluation.rs:251) — same first-match semantics, since keys
// are unique inside one result list.
#pragma once
#include <cstdint>
#include <cstring>
#include <deque>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "msm_kernels.hpp"

namespace h2agg {

// ------------------------------------------------------------------ device side: Fr tape
enum : uint32_t { TAPE_MUL = 0, TAPE_ADD = 1, TAPE_SUB = 2, TAPE_INV = 3,     // INV: dst = 1 / a (b unused); 1 / 0 raises FLAG_DIV_ZERO
                  TAPE_SQRN = 4 };   // dst = a^(2^b), b an IMMEDIATE count (not a register): `pow_constant(x, n)` for n = 2^k
                                     // (verify.rs:498) as ONE operation — k dependent squarings are k levels of the tape otherwise,
                                     // each a round trip through the register file; recorded by the verifier pipeline only
struct TapeOp {
    uint32_t dst, a, b, op;
};
constexpr int REG_WORDS = NL;  // a register = 9 limbs, Montgomery, value < 2r

FP_INLINE Fr reg_load(const uint32_t* regs, uint32_t i) {
    Fr r;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.l[k] = regs[(size_t)i * REG_WORDS + k];
    return r;
}
FP_INLINE void reg_store(uint32_t* regs, uint32_t i, const Fr& v) {
#pragma unroll
    for (int k = 0; k < NL; ++k) regs[(size_t)i * REG_WORDS + k] = v.l[k];
}
// value < 4r -> < 2r
FP_INLINE Fr fr_fold_2r(const Fr& a) {
    int32_t x[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = (int32_t)a.l[i] - (int32_t)km_limb<FrParams>(2, i);
    Fr t = fp_normalize<FrParams>(x);
    const bool neg = (int32_t)t.l[8] < 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = neg ? a.l[i] : t.l[i];
    return r;
}

// one tape operation: MockFieldChip::mul / add / sub  (mock/arith/field.rs:98-105, 39-55); INV = the `b.invert().unwrap()`
// of MockFieldChip::div (:107-114: inversion of zero panics -> FLAG_DIV_ZERO)
FP_INLINE Fr tape_exec(const TapeOp& op, const uint32_t* regs, uint32_t* flags) {
    const Fr a = reg_load(regs, op.a);
    if (op.op == TAPE_INV) {
        if (fp_is_zero_mod<2, FrParams>(a)) atomicOr(flags, FLAG_DIV_ZERO);
        return fp_inv<FrParams>(a);
    }
    if (op.op == TAPE_SQRN) {
        Fr r = a;
#pragma unroll 1
        for (uint32_t i = 0; i < op.b; ++i) r = fp_sqr<FrParams>(r);
        return r;
    }
    const Fr b = reg_load(regs, op.b);
    if (op.op == TAPE_MUL) return fp_mul<FrParams>(a, b);                   // 4/169 + 1 -> < 2r
    if (op.op == TAPE_ADD) return fr_fold_2r(fp_add<FrParams>(a, b));      // < 4r -> < 2r
    return fr_fold_2r(fp_sub<2, FrParams>(a, b));                          // a - b + 2r < 4r -> < 2r
}

// constants: canonical 32-byte integers -> Montgomery registers [0, n)
__global__ void __launch_bounds__(BLOCK) k_tape_load_consts(const uint8_t* __restrict__ consts, uint32_t n,
                                                            uint32_t* __restrict__ regs, uint32_t* flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Fr x = fp_load<FrParams>(consts + 32 * (size_t)i);
    if (!fp_is_canonical<FrParams>(x)) atomicOr(flags, FLAG_NONCANONICAL);
    reg_store(regs, i, fp_to_mont<FrParams>(x));
}
// one dependency level of the tape: every op's inputs were produced by earlier levels
// MockFieldChip::mul / add / sub  (mock/arith/field.rs:98-105, 39-55)
__global__ void __launch_bounds__(BLOCK) k_tape_level(const TapeOp* __restrict__ ops, uint32_t n,
                                                      uint32_t* __restrict__ regs, uint32_t* flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const TapeOp op = ops[i];
    reg_store(regs, op.dst, tape_exec(op, regs, flags));
}
// The whole tape in ONE launch: a single 1024-lane workgroup walks the dependency levels with a barrier
// in between.  Horner chains make tapes deep and narrow (hundreds of levels of a few operations), which is
// launch-bound as one kernel per level; wide levels are strip-mined by the 1024 lanes.
constexpr int TAPE_THREADS = 1024;
__global__ void __launch_bounds__(TAPE_THREADS) k_tape_run(const TapeOp* __restrict__ ops,
                                                           const uint32_t* __restrict__ level_start,
                                                           uint32_t nlevels, uint32_t* __restrict__ regs, uint32_t* flags) {
    // A level is a dependent round trip: operation -> operands -> result -> barrier.  The operation descriptors do not depend
    // on any result, so a lane fetches the one it will run in the NEXT level before it waits at this level's barrier
    // (one memory latency less per level: ~60 levels per aggregation).
    uint32_t lo = nlevels ? level_start[0] : 0, hi = nlevels ? level_start[1] : 0;
    TapeOp nxt = {0, 0, 0, 0};
    if (lo + threadIdx.x < hi) nxt = ops[lo + threadIdx.x];
#pragma unroll 1
    for (uint32_t l = 0; l < nlevels; ++l) {
        const TapeOp first = nxt;
        const uint32_t lo_n = hi, hi_n = l + 1 < nlevels ? level_start[l + 2] : hi;
        if (l + 1 < nlevels && lo_n + threadIdx.x < hi_n) nxt = ops[lo_n + threadIdx.x];
        if (lo + threadIdx.x < hi) reg_store(regs, first.dst, tape_exec(first, regs, flags));
#pragma unroll 1
        for (uint32_t i = lo + threadIdx.x + TAPE_THREADS; i < hi; i += TAPE_THREADS) {
            const TapeOp op = ops[i];
            reg_store(regs, op.dst, tape_exec(op, regs, flags));
        }
        __syncthreads();  // workgroup-scope release/acquire: the next level reads these registers
        lo = lo_n;
        hi = hi_n;
    }
}

// The same walk with the register file in LDS.  A level of k_tape_run is a dependent round trip through
// L2 — results stored, barrier, the next level's operands loaded: ~0.8 us of the ~2.4 us a level takes.  Here every value the
// tape itself reads again lives in an LDS slot for as long as it is live (the host's liveness allocator below hands the slots
// out level by level and reuses them: an aggregation's tape has ~9 500 values, ~150 KiB of LDS hold 4 096), so a level's
// operands are an LDS read behind an LDS-only barrier; constants are converted and placed by this same launch (no separate
// k_tape_load_consts).  Every result still goes to the global register file as well — fire and forget, nothing in here waits
// for those stores — because the consumers behind the tape (k_eval_prep, k_tape_gather) read it from there.
//   op.a / op.b:  bit 31 set: LDS slot (low bits); clear: never happens for a register operand (constants have slots too)
//   op.op:        opcode | (LDS slot of the result, or TAPE_NOSLOT when nothing in the tape reads it) << 8
// A tape whose live values do not fit runs through k_tape_run.
constexpr uint32_t TAPE_LDS_SLOTS = 4096;          // x 36 B = 144 KiB of the CU's 160
constexpr uint32_t TAPE_NOSLOT = 0xffffffu;
constexpr uint32_t TAPE_SLOTBIT = 0x80000000u;
FP_INLINE Fr slot_load(const uint32_t* file, uint32_t s) {
    Fr r;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.l[k] = file[s * NL + k];
    return r;
}
FP_INLINE void slot_store(uint32_t* file, uint32_t s, const Fr& v) {
#pragma unroll
    for (int k = 0; k < NL; ++k) file[s * NL + k] = v.l[k];
}
FP_INLINE void tape_lds_barrier() {
    // LDS traffic only: the global stores of the results are not waited for (their consumers are later kernels)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
FP_INLINE void tape_exec_lds(const TapeOp& op, uint32_t* file, uint32_t* regs, uint32_t* flags) {
    const uint32_t code = op.op & 0xffu, dslot = op.op >> 8;
    const Fr a = slot_load(file, op.a & ~TAPE_SLOTBIT);
    Fr r;
    if (code == TAPE_INV) {
        if (fp_is_zero_mod<2, FrParams>(a)) atomicOr(flags, FLAG_DIV_ZERO);
        r = fp_inv<FrParams>(a);
    } else if (code == TAPE_SQRN) {
        r = a;
#pragma unroll 1
        for (uint32_t i = 0; i < op.b; ++i) r = fp_sqr<FrParams>(r);
    } else {
        const Fr b = slot_load(file, op.b & ~TAPE_SLOTBIT);
        if (code == TAPE_MUL) r = fp_mul<FrParams>(a, b);
        else if (code == TAPE_ADD) r = fr_fold_2r(fp_add<FrParams>(a, b));
        else r = fr_fold_2r(fp_sub<2, FrParams>(a, b));
    }
    if (dslot != TAPE_NOSLOT) slot_store(file, dslot, r);
    reg_store(regs, op.dst, r);
}
__global__ void __launch_bounds__(TAPE_THREADS) k_tape_run_lds(const uint8_t* __restrict__ consts, const uint32_t* __restrict__ cslot,
                                                               uint32_t nconst, const TapeOp* __restrict__ ops,
                                                               const uint32_t* __restrict__ level_start, uint32_t nlevels,
                                                               uint32_t* __restrict__ regs, uint32_t* flags) {
    __shared__ uint32_t file[TAPE_LDS_SLOTS * NL];
    uint32_t lo = nlevels ? level_start[0] : 0, hi = nlevels ? level_start[1] : 0;
    TapeOp nxt = {0, 0, 0, 0};
    if (lo + threadIdx.x < hi) nxt = ops[lo + threadIdx.x];
    uint32_t bad = 0;
#pragma unroll 1
    for (uint32_t i = threadIdx.x; i < nconst; i += TAPE_THREADS) {
        const Fr x = fp_load<FrParams>(consts + 32 * (size_t)i);
        bad |= !fp_is_canonical<FrParams>(x);
        const Fr m = fp_to_mont<FrParams>(x);
        const uint32_t s = cslot[i];
        if (s != TAPE_NOSLOT) slot_store(file, s, m);
        reg_store(regs, i, m);
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    tape_lds_barrier();
#pragma unroll 1
    for (uint32_t l = 0; l < nlevels; ++l) {
        const TapeOp first = nxt;
        const uint32_t lo_n = hi, hi_n = l + 1 < nlevels ? level_start[l + 2] : hi;
        if (l + 1 < nlevels && lo_n + threadIdx.x < hi_n) nxt = ops[lo_n + threadIdx.x];
        if (lo + threadIdx.x < hi) tape_exec_lds(first, file, regs, flags);
#pragma unroll 1
        for (uint32_t i = lo + threadIdx.x + TAPE_THREADS; i < hi; i += TAPE_THREADS) tape_exec_lds(ops[i], file, regs, flags);
        tape_lds_barrier();
        lo = lo_n;
        hi = hi_n;
    }
}

// registers -> canonical 32-byte scalars (MSM scalar buffer / results)
__global__ void __launch_bounds__(BLOCK) k_tape_gather(const uint32_t* __restrict__ regs,
                                                       const uint32_t* __restrict__ idx, uint32_t n,
                                                       uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp_store<FrParams>(out + 32 * (size_t)i, fp_from_mont<FrParams>(reg_load(regs, idx[i])));
}

// Everything between the tape and the sort of an evaluation's (split) multi_exp in ONE launch — it was four (gather,
// bases to Montgomery form, beta * x, GLV decomposition: ~6 us of launch latency each on the evaluation's critical path):
// pair i = (canonical affine point pts[i], tape register idx[i]).  GLV: scal_out receives the glv_decompose() words,
// endo_x the beta * x column; otherwise the canonical scalars.
template <bool GLV>
__global__ void __launch_bounds__(BLOCK) k_eval_prep(const uint32_t* __restrict__ regs, const uint32_t* __restrict__ idx,
                                                     const uint8_t* __restrict__ pts, uint32_t n,
                                                     uint8_t* __restrict__ scal_out, uint8_t* __restrict__ bases_out,
                                                     uint8_t* __restrict__ endo_x, uint32_t* flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t bad = 0;
    const G1Affine p = affine_load_canonical(pts + 64 * (size_t)i, bad);
    affine_store(bases_out + 64 * (size_t)i, p);
    U256 k;
    fp_pack<FrParams>(k.w, fp_from_mont<FrParams>(reg_load(regs, idx[i])));
    if (GLV) {
        Fq beta;
#pragma unroll
        for (int j = 0; j < NL; ++j) beta.l[j] = GlvConst::BETA_MONT[j];
        fp_store<FqParams>(endo_x + 32 * (size_t)i, FQ_MUL(p.x, beta));
        U256 d;
        bad |= !glv_decompose(k, d);
        k = d;
    }
    if (bad) atomicOr(flags, FLAG_NONCANONICAL);
    uint4* q = reinterpret_cast<uint4*>(scal_out + 32 * (size_t)i);
    q[0] = make_uint4(k.w[0], k.w[1], k.w[2], k.w[3]);
    q[1] = make_uint4(k.w[4], k.w[5], k.w[6], k.w[7]);
}

// ------------------------------------------------------------------ host side: AST + symbolic evaluation
// The tape is a DAG of Fr operations; its run time on the device is (number of dependency levels) x (one barrier +
// one multiplication latency), so the recorder keeps it SHALLOW.  Field arithmetic is exact, hence any association
// of the same products / sums yields the identical canonical result; the recorder uses that freedom twice:
//   * multiplicative chains  s_k = s_(k-1) * v  (the scalar pushed down a Horner chain v*acc + q, multiopen.rs:56-60,
//     is v^k after k levels) are recorded as  root * pow(v, k)  with pow() built by halving: depth log2 k, not k;
//   * additive chains  e = e + t_k  (every evaluation merges into the "" entry, evaluation.rs:251-262) stay LAZY
//     (a linked list of terms) until something consumes the sum, then become a balanced tree: depth log2 n, not n.
// Constants are interned by value so that the thousands of `scalar!(v)` leaves share one register.
struct Tape {
    static constexpr uint32_t OPBIT = 0x80000000u;    // provisional id of an operation result (resolved at run time)
    static constexpr uint32_t LAZYBIT = 0x40000000u;  // id of a not-yet-materialised sum
    static constexpr uint32_t NONE = 0xffffffffu;
    std::vector<uint8_t> consts;   // 32 B each; register i < nconst is constant i
    std::vector<TapeOp> ops;       // op k writes register nconst_final + k
    uint32_t nconst = 0;

    struct Key32 {
        uint64_t w[4];
        bool operator==(const Key32& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    };
    struct Key32Hash {
        size_t operator()(const Key32& k) const {
            uint64_t h = k.w[0] * 0x9E3779B97F4A7C15ull;
            h = (h ^ (h >> 29)) + k.w[1] * 0xC2B2AE3D27D4EB4Full;
            h = (h ^ (h >> 31)) + k.w[2] * 0x165667B19E3779F9ull;
            h = (h ^ (h >> 30)) + k.w[3] * 0xD6E8FEB86659FD93ull;
            return (size_t)(h ^ (h >> 32));
        }
    };
    std::unordered_map<Key32, uint32_t, Key32Hash> const_ids;
    struct Chain {
        uint32_t root, c, e;  // register = root * c^e   (root == NONE: just c^e)
    };
    std::vector<Chain> chain_by_op;                   // parallel to ops; c == NONE: the result is not a chain
    std::unordered_map<uint32_t, uint32_t> pow_slot;  // constant -> index into pow_tab
    std::vector<std::vector<uint32_t>> pow_tab;       // pow_tab[slot][e] = register of c^e (NONE: not built yet)
    uint32_t last_pow_c = NONE, last_pow_slot = 0;
    struct Lazy {
        uint32_t parent;  // LAZYBIT id of the sum this one extends, or NONE
        uint32_t head;    // first term when parent == NONE
        uint32_t term;    // the term added here
        uint32_t reg;     // materialised register, or NONE
    };
    std::vector<Lazy> lazies;

    uint32_t add_const(const uint8_t v[32]) {
        Key32 k;
        memcpy(k.w, v, 32);
        auto it = const_ids.find(k);
        if (it != const_ids.end()) return it->second;
        consts.insert(consts.end(), v, v + 32);
        const_ids.emplace(k, nconst);
        return nconst++;
    }
    // A constant whose VALUE arrives later (a transcript challenge still being squeezed while the host records what is done
    // with it): a register of its own, never merged with equal-valued constants; set_const fills it in before the tape runs.
    // Field arithmetic is exact and the recorder only compares register ids, so the results do not depend on when the value
    // becomes known.
    uint32_t add_const_placeholder() {
        consts.insert(consts.end(), 32, (uint8_t)0);
        return nconst++;
    }
    void set_const(uint32_t reg, const uint8_t v[32]) { memcpy(consts.data() + 32 * (size_t)reg, v, 32); }
    static bool is_const(uint32_t r) { return !(r & (OPBIT | LAZYBIT)); }
    uint32_t record(uint32_t opcode, uint32_t a, uint32_t b) {
        a = materialize(a);
        if (opcode != TAPE_SQRN) b = materialize(b);   // (SQRN: b is an immediate count of squarings, never a register id)
        TapeOp o{(uint32_t)ops.size() | OPBIT, a, b, opcode};
        ops.push_back(o);
        chain_by_op.push_back(Chain{NONE, NONE, 0});
        return o.dst;
    }
    const Chain* chain_of(uint32_t r) const {
        if (!(r & OPBIT)) return nullptr;
        const Chain& ch = chain_by_op[r & ~OPBIT];
        return ch.c == NONE ? nullptr : &ch;
    }
    void set_chain(uint32_t r, const Chain& ch) {
        if ((r & OPBIT) && chain_by_op[r & ~OPBIT].c == NONE) chain_by_op[r & ~OPBIT] = ch;
    }
    // a * b where one side is (typically) a `scalar!` leaf: keeps multiplicative chains shallow
    uint32_t mul(uint32_t a, uint32_t b) {
        a = materialize(a);
        b = materialize(b);
        if (is_const(a) && !is_const(b)) std::swap(a, b);  // constant on the right
        if (is_const(b)) {
            if (is_const(a)) {
                if (a == b) return chain(NONE, b, 2);
            } else {
                const Chain* ch = chain_of(a);
                if (ch && ch->c == b) {
                    const Chain cc = *ch;
                    return chain(cc.root, b, cc.e + 1);
                }
            }
            const uint32_t r = record(TAPE_MUL, a, b);
            set_chain(r, Chain{a, b, 1});
            return r;
        }
        return record(TAPE_MUL, a, b);
    }
    uint32_t pow(uint32_t c, uint32_t e) {
        if (e == 1) return c;
        if (c != last_pow_c) {
            auto it = pow_slot.find(c);
            if (it == pow_slot.end()) {
                it = pow_slot.emplace(c, (uint32_t)pow_tab.size()).first;
                pow_tab.emplace_back();
            }
            last_pow_c = c;
            last_pow_slot = it->second;
        }
        const uint32_t slot = last_pow_slot;
        if (pow_tab[slot].size() <= e) pow_tab[slot].resize((size_t)e + 16, NONE);
        if (pow_tab[slot][e] != NONE) return pow_tab[slot][e];
        const uint32_t lo = pow(c, e / 2), hi = pow(c, e - e / 2);   // (same c: the cached slot stays valid)
        const uint32_t r = record(TAPE_MUL, lo, hi);
        pow_tab[slot][e] = r;
        set_chain(r, Chain{NONE, c, e});
        return r;
    }
    uint32_t chain(uint32_t root, uint32_t c, uint32_t e) {
        const uint32_t p = pow(c, e);
        if (root == NONE) return p;
        const uint32_t r = record(TAPE_MUL, root, p);
        set_chain(r, Chain{root, c, e});
        return r;
    }
    // a + b, deferred
    uint32_t add(uint32_t a, uint32_t b) {
        const bool la = (a & LAZYBIT) && lazies[a & ~LAZYBIT].reg == NONE;
        const bool lb = (b & LAZYBIT) && lazies[b & ~LAZYBIT].reg == NONE;
        Lazy z;
        z.reg = NONE;
        if (la) {
            z.parent = a, z.head = NONE, z.term = materialize(b);
        } else if (lb) {
            z.parent = b, z.head = NONE, z.term = materialize(a);
        } else {
            z.parent = NONE, z.head = materialize(a), z.term = materialize(b);
        }
        lazies.push_back(z);
        return (uint32_t)(lazies.size() - 1) | LAZYBIT;
    }
    uint32_t materialize(uint32_t r) {
        if (!(r & LAZYBIT) || r == NONE) return r;
        Lazy& top = lazies[r & ~LAZYBIT];
        if (top.reg != NONE) return top.reg;
        std::vector<uint32_t> terms;
        uint32_t cur = r;
        for (;;) {
            const Lazy& z = lazies[cur & ~LAZYBIT];
            if (z.reg != NONE) {  // an already materialised prefix of the sum
                terms.push_back(z.reg);
                break;
            }
            terms.push_back(z.term);
            if (z.parent == NONE) {
                terms.push_back(z.head);
                break;
            }
            cur = z.parent;
        }
        // oldest term first, then a balanced pairwise tree
        std::vector<uint32_t> lvl(terms.rbegin(), terms.rend());
        while (lvl.size() > 1) {
            size_t o = 0;
            for (size_t i = 0; i + 1 < lvl.size(); i += 2) {
                TapeOp op{(uint32_t)ops.size() | OPBIT, lvl[i], lvl[i + 1], TAPE_ADD};
                ops.push_back(op);
                chain_by_op.push_back(Chain{NONE, NONE, 0});
                lvl[o++] = op.dst;
            }
            if (lvl.size() & 1) lvl[o++] = lvl.back();
            lvl.resize(o);
        }
        lazies[r & ~LAZYBIT].reg = lvl[0];
        return lvl[0];
    }
    uint32_t resolve(uint32_t r) const { return (r & OPBIT) ? nconst + (r & ~OPBIT) : r; }
};

struct SchemaNode {
    enum Kind : uint8_t { COMMITMENT, EVAL, SCALAR, ADD, MUL } kind;
    bool has_commitment;   // own flag (Add/Mul: l.1 || r.1), i.e. what a parent caches in its Box<(_, bool)>
    uint32_t l = 0, r = 0; // children (ADD / MUL)
    int32_t point = -1;    // COMMITMENT: index into Schema::points
    uint32_t reg = 0;      // EVAL / SCALAR: tape constant
    uint32_t key = 0;      // COMMITMENT: interned key (0 = "")
};

struct PreparedEntry {
    uint32_t key;    // interned; 0 = ""
    int32_t point;   // -1 = None
    int64_t scalar;  // -1 = None, otherwise a tape register id (possibly OPBIT / LAZYBIT tagged)
    uint64_t prev = 0;   // the key's stamp before this entry was pushed (see Prepared)
};
// The result lists of eval_prepare live back to back in ONE vector: a call appends its list at the end.  Merging the
// right list of an Add into the left one (evaluation.rs:245-268: entries with an equal key add their scalars, the others
// are appended) is then an in-place compaction, and "is key k in the left list, and where" is one stamp per interned key
// (list id << 32 | position) instead of a hash map per list.  The right subtree is evaluated AFTER the left one and
// overwrites the stamps of the keys it shares with it, so every entry remembers the stamp it displaced (`prev`): for
// the surviving (leftmost) right entry of a key that is exactly the left list's stamp, if the key is there.  List ids are
// never reused, so a stamp of a list that has since been merged away or popped cannot match a live list.
struct Prepared {
    std::vector<PreparedEntry> v;
    std::vector<uint64_t> stamp;   // per key id
    uint32_t next_list = 1;
    void reset(size_t nkeys) {
        v.clear();
        stamp.assign(nkeys, 0);
        next_list = 1;
    }
    uint32_t new_list() { return next_list++; }
    void push(uint32_t list, PreparedEntry e) {
        e.prev = stamp[e.key];
        stamp[e.key] = ((uint64_t)list << 32) | (uint32_t)v.size();
        v.push_back(e);
    }
    void pop() {   // drop the last entry (a temporary one-entry list) and give its key's stamp back
        stamp[v.back().key] = v.back().prev;
        v.pop_back();
    }
};

struct Schema {
    std::vector<SchemaNode> nodes;
    std::vector<uint8_t> points;  // 64 B canonical affine each
    // where every point was copied from (verifier pipeline, while it records an aggregation it will want to run again with
    // other proofs of the same shape: csrc/verifier.inc `AggPlan`)
    bool track_src = false;
    std::vector<const uint8_t*> point_src;
    Tape tape;
    uint32_t one_reg = 0;
    bool has_one = false;
    std::string err;
    std::deque<std::string> key_names{std::string()};                  // interned keys (stable addresses), id 0 = ""
    std::unordered_map<std::string_view, uint32_t> key_ids{{std::string_view(), 0u}};   // views into key_names

    // results of the last eval (names: evaluation.rs:183), as interned key ids
    std::vector<uint32_t> names;
    size_t point_list_len = 0;     // what MockChipCtx::point_list.len() would be after multi_exp

    uint32_t intern(const char* key) {
        const std::string_view sv(key);
        auto it = key_ids.find(sv);
        if (it != key_ids.end()) return it->second;
        const uint32_t id = (uint32_t)key_names.size();
        key_names.emplace_back(sv);
        key_ids.emplace(std::string_view(key_names.back()), id);
        return id;
    }
    void grow_nodes(size_t extra) {
        if (nodes.size() + extra > nodes.capacity()) nodes.reserve(2 * nodes.capacity() + extra);
    }
    uint32_t one() {
        if (!has_one) {
            uint8_t v[32] = {1};
            one_reg = tape.add_const(v);
            has_one = true;
        }
        return one_reg;
    }
    uint32_t add_commitment_id(uint32_t key, const uint8_t p[64]) {
        SchemaNode n;
        n.kind = SchemaNode::COMMITMENT;
        n.has_commitment = true;
        n.key = key;
        n.point = (int32_t)(points.size() / 64);
        points.insert(points.end(), p, p + 64);
        if (track_src) point_src.push_back(p);
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1;
    }
    uint32_t add_commitment(const char* key, const uint8_t p[64]) { return add_commitment_id(intern(key), p); }
    uint32_t add_leaf_scalar(SchemaNode::Kind k, const uint8_t s[32]) {
        SchemaNode n;
        n.kind = k;
        n.has_commitment = false;
        n.reg = tape.add_const(s);
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1;
    }
    // a Scalar / Eval leaf whose value is an existing tape register (a constant, or the result of recorded operations:
    // the verifier pipeline computes evaluation points, expected_h_eval, ... on the tape itself)
    uint32_t add_leaf_reg(SchemaNode::Kind k, uint32_t reg) {
        SchemaNode n;
        n.kind = k;
        n.has_commitment = false;
        n.reg = reg;
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1;
    }
    bool valid(uint32_t id) const { return id < nodes.size(); }
    uint32_t add_binary(SchemaNode::Kind k, uint32_t l, uint32_t r) {
        SchemaNode n;
        n.kind = k;
        n.l = l;
        n.r = r;
        n.has_commitment = nodes[l].has_commitment || nodes[r].has_commitment;  // evaluation.rs:35-36
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1;
    }

    // EvaluationQuery::new (evaluation.rs:100-118): schema = commit!(cq) + eval!(cq)
    uint32_t add_evaluation_query(const char* key, const uint8_t commitment[64], const uint8_t eval[32]) {
        const uint32_t cnode = add_commitment(key, commitment);
        const uint32_t enode = add_leaf_scalar(SchemaNode::EVAL, eval);
        return add_binary(SchemaNode::ADD, cnode, enode);
    }

    // VerifierParams::get_point_schemas + batch_multi_open_proofs (multiopen.rs:23-102).
    // queries: (rotation, evaluation point z, schema node) in VerifierParams::queries order; w: one W
    // commitment per rotation group in first-seen order.  Returns false if the counts differ (the
    // reference's assert_eq!, multiopen.rs:48).
    bool batch_multi_open(const char* key, size_t nq, const int32_t* rotation, const uint8_t* points /*32 B each*/,
                          const uint32_t* qnodes, size_t nw, const uint8_t* w /*64 B each*/, const uint8_t v[32],
                          const uint8_t u[32], uint32_t& w_x_out, uint32_t& w_g_out) {
        return batch_multi_open_impl(key, nq, rotation, points, nullptr, qnodes, nw, w, v, u, 0, 0, w_x_out, w_g_out);
    }
    // the same with the evaluation points and v, u given as tape registers
    bool batch_multi_open_regs(const char* key, size_t nq, const int32_t* rotation, const uint32_t* point_regs,
                               const uint32_t* qnodes, size_t nw, const uint8_t* w, uint32_t v_reg, uint32_t u_reg,
                               uint32_t& w_x_out, uint32_t& w_g_out) {
        return batch_multi_open_impl(key, nq, rotation, nullptr, point_regs, qnodes, nw, w, nullptr, nullptr, v_reg, u_reg,
                                     w_x_out, w_g_out);
    }
    bool batch_multi_open_impl(const char* key, size_t nq, const int32_t* rotation, const uint8_t* points,
                               const uint32_t* point_regs, const uint32_t* qnodes, size_t nw, const uint8_t* w,
                               const uint8_t* v, const uint8_t* u, uint32_t v_reg, uint32_t u_reg, uint32_t& w_x_out,
                               uint32_t& w_g_out) {
        struct Group {
            int32_t rot;
            size_t first;   // index of the group's first query: its evaluation point is the group's
            std::vector<uint32_t> schemas;
        };
        auto leaf_v = [&] { return v ? add_leaf_scalar(SchemaNode::SCALAR, v) : add_leaf_reg(SchemaNode::SCALAR, v_reg); };
        auto leaf_u = [&] { return u ? add_leaf_scalar(SchemaNode::SCALAR, u) : add_leaf_reg(SchemaNode::SCALAR, u_reg); };
        auto leaf_point = [&](size_t i) {
            return points ? add_leaf_scalar(SchemaNode::SCALAR, points + 32 * i) : add_leaf_reg(SchemaNode::SCALAR, point_regs[i]);
        };
        std::vector<Group> groups;
        for (size_t i = 0; i < nq; ++i) {                                     // :33-43
            if (!valid(qnodes[i])) {
                err = "unknown schema node";
                return false;
            }
            size_t g = 0;
            for (; g < groups.size(); ++g)
                if (groups[g].rot == rotation[i]) break;
            if (g == groups.size()) groups.push_back({rotation[i], i, {}});
            groups[g].schemas.push_back(qnodes[i]);
        }
        if (nw != groups.size()) {
            err = "assert_eq!(self.w.len(), points.len()) failed (multiopen.rs:48)";
            return false;
        }
        grow_nodes(3 * nq + 12 * groups.size());
        std::vector<uint32_t> s_of(groups.size());
        for (size_t g = 0; g < groups.size(); ++g) {                          // :56-60  rev().reduce(v*acc + q)
            const std::vector<uint32_t>& sc = groups[g].schemas;
            uint32_t acc = sc.back();
            for (size_t k = sc.size() - 1; k-- > 0;) {
                const uint32_t vn = leaf_v();
                acc = add_binary(SchemaNode::ADD, add_binary(SchemaNode::MUL, vn, acc), sc[k]);
            }
            s_of[g] = acc;
        }
        bool have = false;
        uint32_t w_x = 0, w_g = 0;
        for (size_t gi = groups.size(); gi-- > 0;) {                          // :82-96  enumerate().rev()
            const std::string wkey = std::string(key) + "_w" + std::to_string(gi);
            const uint32_t wid = intern(wkey.c_str());
            const uint32_t cw = add_commitment_id(wid, w + 64 * gi);
            const uint32_t zc = add_binary(SchemaNode::MUL, leaf_point(groups[gi].first), add_commitment_id(wid, w + 64 * gi));
            if (!have) {
                w_x = cw;
                w_g = add_binary(SchemaNode::ADD, zc, s_of[gi]);
                have = true;
            } else {
                w_x = add_binary(SchemaNode::ADD, add_binary(SchemaNode::MUL, leaf_u(), w_x), cw);
                const uint32_t uw = add_binary(SchemaNode::MUL, leaf_u(), w_g);
                w_g = add_binary(SchemaNode::ADD, add_binary(SchemaNode::ADD, uw, zc), s_of[gi]);
            }
        }
        if (!have) {
            err = "no queries";
            return false;
        }
        w_x_out = w_x;
        w_g_out = w_g;
        return true;
    }

    // estimate (evaluation.rs:295-330)
    size_t estimate(uint32_t id, bool scalar) const {
        const SchemaNode& n = nodes[id];
        switch (n.kind) {
        case SchemaNode::COMMITMENT: return 1;
        case SchemaNode::EVAL:
        case SchemaNode::SCALAR: return scalar ? 1 : 0;
        case SchemaNode::ADD:
            if (!nodes[n.l].has_commitment && !nodes[n.r].has_commitment) {
                size_t e = estimate(n.l, false) + estimate(n.r, false);
                return scalar ? e + 1 : e;
            }
            return estimate(n.l, scalar) + estimate(n.r, scalar);
        case SchemaNode::MUL:
            return !nodes[n.l].has_commitment ? estimate(n.r, true) : estimate(n.l, true);
        }
        return 0;
    }

    // eval_prepare (evaluation.rs:205-293); `scalar` = -1 for None.  Appends the result list to out.v and returns its id
    // in `list` (its entries are out.v[start .. out.v.size()), `start` = out.v.size() at the call).  Returns false on the
    // reference's assertion failures (err is set).
    bool eval_prepare(uint32_t id, int64_t scalar, Prepared& out, uint32_t& list) {
        const SchemaNode n = nodes[id];
        switch (n.kind) {
        case SchemaNode::COMMITMENT:                                         // :216-218
            list = out.new_list();
            out.push(list, {n.key, n.point, scalar});
            return true;
        case SchemaNode::EVAL: {                                             // :219-225
            int64_t e = scalar >= 0 ? (int64_t)tape.record(TAPE_MUL, (uint32_t)scalar, n.reg) : (int64_t)n.reg;
            list = out.new_list();
            out.push(list, {0u, -1, e});
            return true;
        }
        case SchemaNode::SCALAR: {                                           // :226-232
            int64_t s = scalar >= 0 ? (int64_t)tape.mul(n.reg, (uint32_t)scalar) : (int64_t)n.reg;
            list = out.new_list();
            out.push(list, {0u, -1, s});
            return true;
        }
        case SchemaNode::ADD: {
            const uint32_t l = n.l, r = n.r;
            const size_t start = out.v.size();
            if (!nodes[l].has_commitment && !nodes[r].has_commitment) {      // :234-244
                uint32_t ll, lr;
                if (!eval_prepare(l, -1, out, ll)) return false;
                const size_t mid = out.v.size();
                if (!eval_prepare(r, -1, out, lr)) return false;
                if (mid - start != 1 || out.v.size() - mid != 1) {
                    err = "assert!(l.len() == 1 && r.len() == 1) failed (evaluation.rs:237-238)";
                    return false;
                }
                int64_t sum = tape.add((uint32_t)out.v[start].scalar, (uint32_t)out.v[mid].scalar);
                if (scalar >= 0) sum = tape.record(TAPE_MUL, (uint32_t)scalar, (uint32_t)sum);
                out.pop();
                out.pop();
                list = out.new_list();
                out.push(list, {0u, -1, sum});
                return true;
            }
            // :245-268  merge entries with equal key by adding their scalars (None == one).
            // Every list eval_prepare returns has unique keys (a single entry, or the `res` of an Add), so pushing the
            // LEFT entries through the reference's find-or-push loop never merges anything: the left list is taken
            // over as `res`, and only the right entries are looked up.  Same result and order as the reference,
            // without its O(n^2) scan.
            uint32_t res, rhs;
            if (!eval_prepare(l, scalar, out, res)) return false;
            const size_t mid = out.v.size();
            if (!eval_prepare(r, scalar, out, rhs)) return false;
            const size_t end = out.v.size();
            size_t w = mid;                                                  // right entries kept are compacted to here
            for (size_t k = mid; k < end; ++k) {
                const PreparedEntry ev = out.v[k];
                if ((uint32_t)(ev.prev >> 32) == res) {                       // the key is in the left list
                    const size_t pos = (uint32_t)ev.prev;
                    PreparedEntry& p = out.v[pos];
                    const uint32_t a = p.scalar >= 0 ? (uint32_t)p.scalar : one();
                    const uint32_t b = ev.scalar >= 0 ? (uint32_t)ev.scalar : one();
                    p.scalar = tape.add(a, b);
                    out.stamp[ev.key] = ev.prev;
                } else {
                    out.v[w] = ev;
                    out.stamp[ev.key] = ((uint64_t)res << 32) | (uint32_t)w;
                    ++w;
                }
            }
            out.v.resize(w);
            list = res;
            return true;
        }
        case SchemaNode::MUL: {                                              // :271-291
            const uint32_t l = n.l, r = n.r;
            const size_t start = out.v.size();
            uint32_t ls;
            uint32_t rem;
            if (!nodes[l].has_commitment) {
                if (!eval_prepare(l, -1, out, ls)) return false;
                rem = r;
            } else {
                if (!eval_prepare(r, -1, out, ls)) return false;
                rem = l;
            }
            if (out.v.size() - start != 1) {
                err = "assert_eq!(s.len(), 1) failed (evaluation.rs:282)";
                return false;
            }
            int64_t sv = out.v[start].scalar;
            if (sv < 0) {   // a bare Commitment on the "scalar" side: the reference unwraps None
                err = "Mul: the scalar side carries no scalar (reference: `s.unwrap()` on None, evaluation.rs:284-288)";
                return false;
            }
            out.pop();
            if (scalar >= 0) sv = tape.mul((uint32_t)scalar, (uint32_t)sv);
            return eval_prepare(rem, sv, out, list);
        }
        }
        return false;
    }
    bool eval_prepare(uint32_t id, int64_t scalar, Prepared& out) {
        out.reset(key_names.size());
        uint32_t list;
        return eval_prepare(id, scalar, out, list);
    }
};

}  // namespace h2agg
