#!/usr/bin/env python3
"""Fixed-base levels (h2agg_bases_precompute) against the ordinary path: single MSM latency, back-to-back rate, 16-MSM batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
for lg in (12, 14, 16, 17, 18):
    n = (1 << lg) - 6
    k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8); k[:, 31] &= 0x1f
    table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
    B = 16 if lg <= 17 else 2
    s = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8); s[:, :, 31] &= 0x1f
    d = s.to(dev); out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
    res = {}
    for mode in ("ordinary", "fixed-base"):
        if mode == "fixed-base":
            t0 = time.perf_counter(); eng.bases_precompute(table, 0); tp = time.perf_counter() - t0
        for ovl in (0, 2):
            eng.msm_set_tail_overlap(ovl)
            eng.g1_msm_device(table, d[0].data_ptr(), n)
            t0 = time.perf_counter()
            for _ in range(10):
                if ovl: eng.g1_msm_device_async(table, d[0].data_ptr(), n, out[0].data_ptr())
                else: eng.g1_msm_device(table, d[0].data_ptr(), n)
            eng.synchronize()
            res[(mode, ovl)] = (time.perf_counter() - t0) / 10 * 1e3
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr())
        eng.synchronize()
        res[(mode, "batch")] = (time.perf_counter() - t0) / 5 * 1e3
        eng.msm_set_tail_overlap(0)
    print("2^%d: single latency %.3f -> %.3f ms | back to back %.3f -> %.3f ms | batch of %d: %.3f -> %.3f ms | precompute %.1f ms"
          % (lg, res[("ordinary", 0)], res[("fixed-base", 0)], res[("ordinary", 2)], res[("fixed-base", 2)], B,
             res[("ordinary", "batch")], res[("fixed-base", "batch")], tp * 1e3), flush=True)
    eng.bases_free(table)
