#!/usr/bin/env python3
"""host sponge alone: one proof's element stream (1 088 elements = 136 permutations at rate 8), one thread, both kernels;
no GPU.   python tools/sponge_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
nelem = 8 * 136
elems = b"".join(((0x9E3779B97F4A7C15 * (i + 1)) ** 3 % R).to_bytes(32, "little") for i in range(nelem))
upto = [nelem]
for kern in ("scalar", "ifma"):
    if kern == "ifma" and pkg.host_sponge_kind() != "ifma":
        continue
    pkg.poseidon_squeeze_batch_host(elems, 1, upto, 1, kern)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); out = pkg.poseidon_squeeze_batch_host(elems, 1, upto, 1, kern); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%-6s  %.1f us per stream of %d permutations = %.2f us per permutation   (%s)" % (kern, ts[len(ts) // 2] * 1e6, nelem // 8 + 1, ts[len(ts) // 2] * 1e6 / (nelem // 8 + 1), out[:8].hex()))
