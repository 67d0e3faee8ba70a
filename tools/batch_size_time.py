#!/usr/bin/env python3
"""Fixed-base batched instance-commitment MSMs (2^17 - 6 scalars each, as in the aggregation leg): time per batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n = (1 << lg) - 6
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
for mode in ("ordinary", "fixed-base"):
    if mode == "fixed-base":
        eng.bases_precompute(table, 0)
    for B in (4, 16, 32, 64, 128, 256):
        s = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8); s[:, :, 31] &= 0x1f
        d = s.to(dev); out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr())
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%-10s 2^%d x %3d: %8.3f ms  %6.1f M points/s" % (mode, lg, B, dt * 1e3, B * n / dt / 1e6), flush=True)
