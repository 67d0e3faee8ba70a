#!/usr/bin/env python3
"""Generates csrc/fp_asm.inc: Montgomery products for gfx950 as ONE inline-asm block each, registers chosen by hand.

Why: k_msm_accumulate's register pressure.  Written in C++ (fp.hpp's chains), a mixed addition keeps ~166 VGPRs alive because
LLVM interleaves independent products; at 166 the kernel runs 3 waves per SIMD with no room beside it.  A product as one opaque
block has a known footprint — its operands plus the fixed temporaries below — and cannot be interleaved with anything, so the
whole bucket insertion fits 128 VGPRs with no scratch.

Fixed temporaries (clobbered by the blocks that use them; everything else is an operand the compiler places):
    chain 0:  t v[126:127]   quotient digits m v115..v123
    chain 1:  t v[124:125]   quotient digits m v106..v114
A squaring keeps its doubled operand 2*a_i (i < 8) in the register of RESULT limb i until that limb is produced: the last
column that reads 2*a_i is i + 8, result limb i is written in column i + 9.
Single-chain blocks use chain 0's registers only; the two-partial-sum product uses both t's and chain 0's digits.
All products: product scanning, R = 2^261, 9 x 29-bit limbs, bounds as in fp.hpp.

Hazards: none to pad.  A v_mad_u64_u32 result is interlocked like any VALU result (hipcc itself emits dependent multiply-adds
back to back in plain C++ code); the `s_nop 0` it puts behind every multiply-add of fp.hpp's fenced chains comes from the
empty `asm("" : "+v"(t))` fence there (the hazard recogniser assumes the worst about an inline-asm reader), not from the
hardware.  The blocks write vcc (carry-out, unused) and never read it.  PAD_MADS = True brings the wait states back for A/B.

    python3 tools/gen_fp_asm.py > halo2-snark-aggregator_amd/csrc/fp_asm.inc
"""
NL = 9
MASK = "0x1fffffff"
PAD_MADS = False


class Chain:
    def __init__(self, t_lo, m_base):
        self.lo, self.hi = f"v{t_lo}", f"v{t_lo + 1}"
        self.t = f"v[{t_lo}:{t_lo + 1}]"
        self.m = [f"v{m_base + i}" for i in range(NL)]
        self.clob_t = [self.lo, self.hi] + self.m


C0 = Chain(126, 115)
C1 = Chain(124, 106)


class Block:
    """instruction list with the one hazard rule applied on emission"""

    def __init__(self):
        self.lines = []
        self.last_mad_dst = None   # t register written by the instruction just emitted, if it was a v_mad_u64_u32

    def emit(self, text, reads=(), mad_dst=None):
        if PAD_MADS and self.last_mad_dst is not None and self.last_mad_dst in reads:
            self.lines.append("s_nop 0")
        self.lines.append(text)
        self.last_mad_dst = mad_dst

    def mad(self, ch, x, y, first=False):
        self.emit(f"v_mad_u64_u32 {ch.t}, vcc, {x}, {y}, {'0' if first else ch.t}", reads=() if first else (ch.t,), mad_dst=ch.t)


def rr_interleave(lists):
    """round-robin merge of per-chain instruction thunks"""
    out = []
    n = max(len(l) for l in lists)
    for i in range(n):
        for l in lists:
            if i < len(l):
                out.append(l[i])
    return out


def montgomery(b, chains, columns, outs, mod, ninv):
    """chains[c]: Chain; columns[c][k]: (x, y) product terms of column k of product c; outs[c][j]: result limb registers.
    The products run in lock step, their multiply-adds written alternately."""
    nc = len(chains)
    for k in range(2 * NL - 1):
        per = []
        for c in range(nc):
            ch = chains[c]
            terms = list(columns[c][k])
            for i in range(NL):
                j = k - i
                if i < k and 0 <= j < NL:
                    terms.append((ch.m[i], mod[j]))
            per.append([(lambda ch=ch, x=x, y=y, f=(k == 0 and n == 0): b.mad(ch, x, y, first=f)) for n, (x, y) in enumerate(terms)])
        for th in rr_interleave(per):
            th()
        if k < NL:
            for ch in chains:
                b.emit(f"v_mul_lo_u32 {ch.m[k]}, {ch.lo}, {ninv}", reads=(ch.t,))
            for ch in chains:
                b.emit(f"v_and_b32 {ch.m[k]}, {MASK}, {ch.m[k]}")
            for ch in chains:
                b.mad(ch, ch.m[k], mod[0])
            for ch in chains:
                b.emit(f"v_lshrrev_b64 {ch.t}, 29, {ch.t}", reads=(ch.t,))
        elif k < 2 * NL - 2:
            for c, ch in enumerate(chains):
                b.emit(f"v_and_b32 {outs[c][k - NL]}, {MASK}, {ch.lo}", reads=(ch.t,))
            for ch in chains:
                b.emit(f"v_lshrrev_b64 {ch.t}, 29, {ch.t}", reads=(ch.t,))
        else:
            for c, ch in enumerate(chains):
                b.emit(f"v_and_b32 {outs[c][k - NL]}, {MASK}, {ch.lo}", reads=(ch.t,))
            for c, ch in enumerate(chains):
                b.emit(f"v_alignbit_b32 {outs[c][NL - 1]}, {ch.hi}, {ch.lo}, 29", reads=(ch.t,))


def montgomery_2sum(b, cols_a, cols_b, out, mod, ninv):
    """ONE product-sum (a*b + c*d, one reduction) whose column terms are split over two partial sums: chain 0's t carries on from
    the previous column, chain 1's t starts every column at 0 and is folded in before the quotient digit (v_lshl_add_u64)."""
    A, B = C0, C1
    for k in range(2 * NL - 1):
        ta = list(cols_a[k])
        tb = list(cols_b[k])
        red = [(A.m[i], mod[k - i]) for i in range(NL) if i < k and 0 <= k - i < NL]
        # balance the two partial sums: reduction terms go to the shorter list
        for r in red:
            (ta if len(ta) <= len(tb) else tb).append(r)
        la = [(lambda x=x, y=y, f=(k == 0 and n == 0): b.mad(A, x, y, first=f)) for n, (x, y) in enumerate(ta)]
        lb = [(lambda x=x, y=y, f=(n == 0): b.mad(B, x, y, first=f)) for n, (x, y) in enumerate(tb)]
        for th in rr_interleave([la, lb]):
            th()
        b.emit(f"v_lshl_add_u64 {A.t}, {B.t}, 0, {A.t}", reads=(A.t, B.t))
        if k < NL:
            b.emit(f"v_mul_lo_u32 {A.m[k]}, {A.lo}, {ninv}", reads=(A.t,))
            b.emit(f"v_and_b32 {A.m[k]}, {MASK}, {A.m[k]}")
            b.mad(A, A.m[k], mod[0])
            b.emit(f"v_lshrrev_b64 {A.t}, 29, {A.t}", reads=(A.t,))
        elif k < 2 * NL - 2:
            b.emit(f"v_and_b32 {out[k - NL]}, {MASK}, {A.lo}", reads=(A.t,))
            b.emit(f"v_lshrrev_b64 {A.t}, 29, {A.t}", reads=(A.t,))
        else:
            b.emit(f"v_and_b32 {out[k - NL]}, {MASK}, {A.lo}", reads=(A.t,))
            b.emit(f"v_alignbit_b32 {out[NL - 1]}, {A.hi}, {A.lo}, 29", reads=(A.t,))


def cols_mul(a, bb):
    return [[(a[i], bb[k - i]) for i in range(NL) if 0 <= k - i < NL] for k in range(2 * NL - 1)]


def cols_sqr(a, d):
    cols = []
    for k in range(2 * NL - 1):
        t = []
        for i in range(NL):
            j = k - i
            if j < i or j >= NL:
                continue
            t.append((a[i], a[i]) if i == j else (d[i], a[j]))
        cols.append(t)
    return cols


def ops(base, n=NL):
    return [f"%{base + i}" for i in range(n)]


def cxx(name, doc, params, body_lines, outs, ins, clob):
    text = "\\n\\t".join(body_lines)
    o = ", ".join(outs)
    i = ", ".join(ins)
    c = ", ".join(f'"{r}"' for r in clob + ["vcc"])
    return (f"// {doc}\ntemplate <class P>\nFP_INLINE void {name}({params}) {{\n"
            f"    asm(\"{text}\"\n        : {o}\n        : {i}\n        : {c});\n}}\n")


def limbs(var, constraint):
    return [f'"{constraint}"({var}.l[{i}])' for i in range(NL)]


def consts():
    return [f'"s"(P::MOD[{i}])' for i in range(NL)] + ['"s"(P::NINV)']


def gen():
    out = ["// GENERATED by tools/gen_fp_asm.py — do not edit.  See that file for the register layout and the hazard rule.\n"]
    # ---- single chain
    b = Block()
    montgomery(b, [C0], [cols_mul(ops(0), ops(9))], [ops(0)], ops(18), "%27")
    out.append(cxx("fpa_mul_ip", "a <- a*b / 2^261 mod m (in place: limb j of the result is written after the last read of a.l[j]).  Bounds as fp_mul.",
                   "Fp<P>& a, const Fp<P>& b", b.lines, limbs("a", "+v"), limbs("b", "v") + consts(), C0.clob_t))
    b = Block()
    montgomery(b, [C0], [cols_mul(ops(9), ops(18))], [ops(0)], ops(27), "%36")
    out.append(cxx("fpa_mul", "r <- a*b / 2^261 mod m.", "Fp<P>& r, const Fp<P>& a, const Fp<P>& b", b.lines,
                   limbs("r", "=&v"), limbs("a", "v") + limbs("b", "v") + consts(), C0.clob_t))
    b = Block()
    a, r = ops(9), ops(0)
    for i in range(NL - 1):
        b.emit(f"v_lshlrev_b32 {r[i]}, 1, {a[i]}")
    montgomery(b, [C0], [cols_sqr(a, r)], [r], ops(18), "%27")
    out.append(cxx("fpa_sqr", "r <- a^2 / 2^261 mod m: cross terms once, with a doubled operand (< 2^30).", "Fp<P>& r, const Fp<P>& a",
                   b.lines, limbs("r", "=&v"), limbs("a", "v") + consts(), C0.clob_t))
    b = Block()
    montgomery(b, [C0], [[x + y for x, y in zip(cols_mul(ops(0), ops(9)), cols_mul(ops(18), ops(27)))]], [ops(0)], ops(36), "%45")
    out.append(cxx("fpa_mul2_ip1", "a <- (a*b + c*d) / 2^261 mod m with ONE reduction, one chain.",
                   "Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d", b.lines, limbs("a", "+v"),
                   limbs("b", "v") + limbs("c", "v") + limbs("d", "v") + consts(), C0.clob_t))
    # ---- two products in lock step
    b = Block()
    montgomery(b, [C0, C1], [cols_mul(ops(0), ops(18)), cols_mul(ops(9), ops(27))], [ops(0), ops(9)], ops(36), "%45")
    out.append(cxx("fpa_mul_dual_ip", "(a, c) <- (a*b, c*d), both in place, multiply-adds of the two chains written alternately (no wait states).",
                   "Fp<P>& a, const Fp<P>& b, Fp<P>& c, const Fp<P>& d", b.lines, limbs("a", "+v") + limbs("c", "+v"),
                   limbs("b", "v") + limbs("d", "v") + consts(), C0.clob_t + C1.clob_t))
    b = Block()
    a, c, r0, r1 = ops(18), ops(27), ops(0), ops(9)
    for i in range(NL - 1):
        b.emit(f"v_lshlrev_b32 {r0[i]}, 1, {a[i]}")
        b.emit(f"v_lshlrev_b32 {r1[i]}, 1, {c[i]}")
    montgomery(b, [C0, C1], [cols_sqr(a, r0), cols_sqr(c, r1)], [r0, r1], ops(36), "%45")
    out.append(cxx("fpa_sqr_dual", "(r0, r1) <- (a^2, c^2), two chains in lock step.", "Fp<P>& r0, Fp<P>& r1, const Fp<P>& a, const Fp<P>& c",
                   b.lines, limbs("r0", "=&v") + limbs("r1", "=&v"), limbs("a", "v") + limbs("c", "v") + consts(),
                   C0.clob_t + C1.clob_t))
    # ---- a*b + c*d, one reduction, two partial sums per column
    b = Block()
    montgomery_2sum(b, cols_mul(ops(0), ops(9)), cols_mul(ops(18), ops(27)), ops(0), ops(36), "%45")
    out.append(cxx("fpa_mul2_ip", "a <- (a*b + c*d) / 2^261 mod m with ONE reduction (27 products of < 2^58 per column fit 64 bits); two partial sums per column, folded before the quotient digit.",
                   "Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d", b.lines, limbs("a", "+v"),
                   limbs("b", "v") + limbs("c", "v") + limbs("d", "v") + consts(), C0.clob_t + [C1.lo, C1.hi]))
    return "\n".join(out)


if __name__ == "__main__":
    print(gen())
