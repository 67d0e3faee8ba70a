// CPU check of the LDS register-file allocator of the Fr tape (csrc/schema_api.inc: schedule_levels + tape_lds_assign), which
// tests/test_tape_lds_alloc.py extracts from the product source into tape_lds_extract.inc next to this file's build.
// The kernel (csrc/schema.hpp k_tape_run_lds) is simulated with the semantics the hardware gives it: inside a level every lane
// reads its operands' slots and writes its result's slot in no particular order, so a slot written in a level must not be read
// or written by any other operation of that level; between levels there is a barrier.  Values are integers mod 2^61 - 1.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace h2agg {
enum : uint32_t { TAPE_MUL = 0, TAPE_ADD = 1, TAPE_SUB = 2, TAPE_INV = 3, TAPE_SQRN = 4 };
struct TapeOp { uint32_t dst, a, b, op; };
constexpr uint32_t TAPE_LDS_SLOTS = H2AGG_TAPE_LDS_SLOTS, TAPE_NOSLOT = H2AGG_TAPE_NOSLOT, TAPE_SLOTBIT = 0x80000000u;
#include "tape_lds_extract.inc"
}
using namespace h2agg;

static const uint64_t P = (1ull << 61) - 1;
static uint64_t mulm(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P); }
static uint64_t exec(uint32_t op, uint64_t a, uint64_t b, uint32_t imm) {
    if (op == TAPE_MUL) return mulm(a, b);
    if (op == TAPE_ADD) return (a + b) % P;
    if (op == TAPE_SUB) return (a + P - b) % P;
    if (op == TAPE_INV) return mulm(a, 0x1234567ull);   // (any unary function)
    uint64_t r = a;
    for (uint32_t i = 0; i < imm; ++i) r = mulm(r, r);
    return r;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// kind: 0 mixed, 1 one wide level, 2 a chain that recycles, 3 too many live, 4 constants first read at the end
static int run(int kind, uint32_t nconst, uint32_t nops, bool expect_fit) {
    std::vector<uint64_t> want(nconst + nops);
    for (uint32_t i = 0; i < nconst; ++i) want[i] = rnd() % P;
    std::vector<TapeOp> ops(nops);
    for (uint32_t k = 0; k < nops; ++k) {
        const uint32_t hi = nconst + k;
        uint32_t op = (uint32_t)(rnd() % 3), a, b;
        if (kind == 0) {
            a = (k % 3 == 0 && k) ? hi - 1 : (uint32_t)(rnd() % hi);
            b = (uint32_t)(rnd() % hi);
            if (k % 97 == 5) { op = TAPE_INV; b = a; }
            if (k % 101 == 7) { op = TAPE_SQRN; b = (uint32_t)(rnd() % 9); }
        } else if (kind == 1) {
            a = (uint32_t)(rnd() % nconst); b = (uint32_t)(rnd() % nconst);
        } else if (kind == 2) {
            a = hi - 1; b = k > 1 ? hi - 2 : (uint32_t)(rnd() % nconst);
        } else if (kind == 3) {
            a = k ? hi - 1 : 0; b = k % nconst;
        } else {
            a = k ? hi - 1 : 0; b = nconst - 1 - (k % nconst);
        }
        ops[k] = TapeOp{hi, a, b, op};
        want[hi] = exec(op, want[a], op == TAPE_SQRN ? 0 : want[b], b);
    }
    std::vector<TapeOp> sorted;
    std::vector<uint32_t> lstart, cslot;
    uint32_t maxlevel = 0, peak = 0;
    if (!schedule_levels(ops, nconst + nops, sorted, lstart, maxlevel)) { printf("schedule_levels refused\n"); return 1; }
    const std::vector<TapeOp> plain = sorted;
    const bool fit = tape_lds_assign(sorted, lstart, nconst, nconst + nops, cslot, &peak);
    if (fit != expect_fit) { printf("kind %d: fit = %d, expected %d (peak %u)\n", kind, (int)fit, (int)expect_fit, peak); return 1; }
    if (!fit) {
        for (size_t i = 0; i < sorted.size(); ++i)
            if (sorted[i].a != plain[i].a || sorted[i].b != plain[i].b || sorted[i].op != plain[i].op) { printf("refused tape was modified\n"); return 1; }
        printf("kind %d: %u consts, %u ops, %u levels, peak %u > %u slots: refused, tape untouched\n", kind, nconst, nops, maxlevel, peak, TAPE_LDS_SLOTS);
        return 0;
    }
    if (peak > TAPE_LDS_SLOTS) { printf("peak %u beyond the file\n", peak); return 1; }
    std::vector<uint64_t> file(TAPE_LDS_SLOTS, 0xdeadbeefull), regs(nconst + nops, 0);
    std::vector<uint32_t> stamp_w(TAPE_LDS_SLOTS, 0xffffffffu), stamp_r(TAPE_LDS_SLOTS, 0xffffffffu);
    for (uint32_t i = 0; i < nconst; ++i) {
        regs[i] = want[i];
        if (cslot[i] != TAPE_NOSLOT) {
            if (cslot[i] >= TAPE_LDS_SLOTS || stamp_w[cslot[i]] == 0) { printf("constant slots collide\n"); return 1; }
            stamp_w[cslot[i]] = 0;
            file[cslot[i]] = want[i];
        }
    }
    for (uint32_t l = 0; l + 1 < lstart.size(); ++l) {
        std::vector<uint64_t> res(lstart[l + 1] - lstart[l]);
        for (uint32_t k = lstart[l]; k < lstart[l + 1]; ++k) {       // all reads of the level (against the state the barrier left)
            const TapeOp& o = sorted[k];
            const uint32_t code = o.op & 0xffu;
            if (!(o.a & TAPE_SLOTBIT) || (code != TAPE_SQRN && !(o.b & TAPE_SLOTBIT))) { printf("operand without a slot\n"); return 1; }
            const uint32_t sa = o.a & ~TAPE_SLOTBIT, sb = o.b & ~TAPE_SLOTBIT;
            if (sa >= TAPE_LDS_SLOTS || (code != TAPE_SQRN && sb >= TAPE_LDS_SLOTS)) { printf("slot out of range\n"); return 1; }
            stamp_r[sa] = l + 1;
            if (code != TAPE_SQRN) stamp_r[sb] = l + 1;
            res[k - lstart[l]] = exec(code, file[sa], code == TAPE_SQRN ? 0 : file[sb], o.b);
        }
        for (uint32_t k = lstart[l]; k < lstart[l + 1]; ++k) {       // all writes
            const TapeOp& o = sorted[k];
            const uint32_t ds = o.op >> 8;
            regs[o.dst] = res[k - lstart[l]];
            if (ds == TAPE_NOSLOT) continue;
            if (ds >= TAPE_LDS_SLOTS) { printf("result slot out of range\n"); return 1; }
            if (stamp_r[ds] == l + 1) { printf("level %u writes slot %u that the same level reads\n", l + 1, ds); return 1; }
            if (stamp_w[ds] == l + 1) { printf("level %u writes slot %u twice\n", l + 1, ds); return 1; }
            stamp_w[ds] = l + 1;
            file[ds] = res[k - lstart[l]];
        }
    }
    for (uint32_t r = 0; r < nconst + nops; ++r)
        if (regs[r] != want[r]) { printf("kind %d: register %u differs\n", kind, r); return 1; }
    printf("kind %d: %u consts, %u ops, %u levels, peak %u slots: every register right, no hazard inside a level\n", kind, nconst, nops, maxlevel, peak);
    return 0;
}

int main() {
    int bad = 0;
    bad |= run(0, 300, 6000, true);
    bad |= run(0, 1500, 8000, true);
    bad |= run(1, 3000, 1000, true);
    bad |= run(1, 3000, 3500, true);       // one level wider than the workgroup; results nothing reads take no slot
    bad |= run(2, 5, 200000, true);
    bad |= run(3, 5000, 5001, false);
    bad |= run(4, 3900, 3900, true);
    bad |= run(4, TAPE_LDS_SLOTS - 2, 5000, true);    // constants + the chain's two live results = exactly the file
    bad |= run(4, TAPE_LDS_SLOTS - 1, 5000, false);
    printf(bad ? "TAPE-LDS-ALLOC-FAILED\n" : "TAPE-LDS-ALLOC-OK\n");
    return bad;
}
