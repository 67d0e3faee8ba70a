#!/usr/bin/env python3
"""Timing of the batched scalar-mul entry points (round-1 verdict item 9): h2agg_g1_batch_scalar_mul at n = 256 (latency-
shaped: ~0.5 ms chain) and h2agg_bases_generate at 2^20 (fixed-base comb).  H2AGG_SCALAR_MUL=ladder selects the round-1
bit-serial kernels for comparison.   python tools/scalar_mul_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as entry

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
pkg = entry.load_package()
eng = pkg.H2Agg(0)
rng = np.random.Generator(np.random.PCG64(9))
g = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")


def frs(n):
    raw = rng.bytes(64 * n)
    return b"".join((int.from_bytes(raw[64 * i:64 * i + 64], "little") % R).to_bytes(32, "little") for i in range(n))


for n in (1, 256, 4096, 65536):
    bases = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(g * n, frs(n)))
    sc = frs(n)
    eng.g1_batch_scalar_mul(bases, sc)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        eng.g1_batch_scalar_mul(bases, sc)
    print("g1_batch_scalar_mul n=%6d  %.3f ms per call (host buffers, synchronous)" % (n, (time.perf_counter() - t0) / reps * 1e3))
for log2n in (17, 20):
    n = 1 << log2n
    d_k = torch.frombuffer(bytearray(frs(n)), dtype=torch.uint8).cuda()
    h = eng.bases_generate(d_k.data_ptr(), n)       # first call also builds the comb table
    eng.bases_free(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = eng.bases_generate(d_k.data_ptr(), n)
    print("bases_generate 2^%d  %.3f ms" % (log2n, (time.perf_counter() - t0) * 1e3))
    eng.bases_free(h)
