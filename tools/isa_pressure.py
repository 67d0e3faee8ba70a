#!/usr/bin/env python3
"""Rough VGPR liveness over a straight-line range of a gfx950 .s file (lines lo..hi): backward scan, every register operand
in the first position of a non-store instruction is a def, the rest are uses.  Prints the live count at sched barriers and
the peak.  A tool for finding WHERE a kernel's register pressure peaks, not an allocator."""
import re, sys
path, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = open(path).read().split('\n')[lo - 1:hi]
RE = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
def regs(tok):
    out = set()
    for m in RE.finditer(tok):
        if m.group(1) is not None: out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
ins = []
for n, l in enumerate(lines):
    s = l.split(';')[0].strip()
    if not s or s.startswith('.') or s.endswith(':'):
        if 'sched_barrier' in l: ins.append((n, 'SB', set(), set()))
        continue
    op, _, rest = s.partition(' ')
    ops = [o.strip() for o in rest.split(',')]
    if op.startswith(('global_store', 'scratch_store', 'ds_write', 'buffer_store', 's_', 'v_cmp')) or op.startswith('global_load_lds'):
        d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
    else:
        d = regs(ops[0]) if ops else set()
        u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    ins.append((n, op, d, u))
live = set(); peak = 0; peakn = 0; out = []
for n, op, d, u in reversed(ins):
    live -= d; live |= u
    if len(live) > peak: peak, peakn = len(live), n
    if op == 'SB': out.append((n + lo, len(live)))
for n, c in reversed(out): print(f'line {n}: live {c}')
print(f'peak {peak} at line {peakn + lo}')
