#!/usr/bin/env python3
"""Timing of pathological scalar distributions at 2^20 points (correctness of these is covered by the tests; this looks
for performance cliffs): uniform, all equal, all ones, 16-bit scalars, half zero."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
n = 1 << 20
rng = np.random.Generator(np.random.PCG64(3))
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 31] &= 0x1f
table = eng.bases_generate(torch.from_numpy(k).to(dev).data_ptr(), n)
def case(name, arr):
    d = torch.from_numpy(arr).to(dev)
    for glv in (-1, 1):
        eng.msm_configure_glv(glv)
        eng.g1_msm_device(table, d.data_ptr(), n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): eng.g1_msm_device(table, d.data_ptr(), n)
        print("%-14s %-5s %8.2f ms" % (name, "glv" if glv == 1 else "plain", (time.perf_counter() - t0) / 3 * 1e3), flush=True)
u = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); u[:, 31] &= 0x1f
case("uniform", u)
e = np.tile(u[0], (n, 1)); case("all equal", e)
o = np.zeros((n, 32), np.uint8); o[:, 0] = 1; case("all ones", o)
s = np.zeros((n, 32), np.uint8); s[:, :2] = u[:, :2]; case("16-bit", s)
h = u.copy(); h[::2] = 0; case("half zero", h)
m = np.tile(((21888242871839275222246405745257275088548364400416034343698204186575808495617 - 1).to_bytes(32, "little")), n)
case("all r-1", np.frombuffer(m, np.uint8).reshape(n, 32).copy())
