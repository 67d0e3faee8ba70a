#!/usr/bin/env python3
"""host-side pairing check of an aggregation (h2agg_pairing_check on two pairs with repeating G2 points: csrc/pairing.hpp), timed
through the C ABI with ctx = NULL — runs wherever the library loads, no GPU.   python tools/pairing_time.py [reps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
from oracle import bn254 as O
from oracle import pairing as E
pkg = entry.load_package(); lib = pkg.load_library()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
def g2b(q): return b"".join(O.fe_to_bytes(v) for v in (q[0][0], q[0][1], q[1][0], q[1][1]))
s = 0x1234567
sg2 = E.g2_mul(s, E.G2)
ts = []
ok = C.c_int(-1)
for i in range(reps):
    a = 1000003 * (i + 1) % O.R
    # e(a s G, G2) * e(-a G, s G2) = 1
    g1 = O.aff_to_bytes(O.scalar_mul(a * s % O.R, O.G1)) + O.aff_to_bytes(O.scalar_mul((-a) % O.R, O.G1))
    g2 = g2b(E.G2) + g2b(sg2)
    t0 = time.perf_counter()
    rc = lib.h2agg_pairing_check(None, g1, g2, 2, C.byref(ok))
    ts.append(time.perf_counter() - t0)
    assert rc == 0 and ok.value == 1, (rc, ok.value)
ts = sorted(ts[5:])
print("pairing check of 2 pairs (prepared G2 lines): median %.1f us, min %.1f us over %d calls" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6, len(ts)))
