#!/usr/bin/env python3
"""16 MSMs over one 2^lg-point table (the instance columns of BASELINE.json configs[4]'s per-GPU share): ordinary path against
fixed-base levels (h2agg_bases_precompute), results compared.    python tools/fixed_base_big.py [lg ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
ORD_ONLY = "--ordinary-only" in sys.argv      # (for counter passes: one path only)
FIX_ONLY = "--fixed-only" in sys.argv          # ... the path BASELINE.json configs[4]'s share runs on since round 5
for lg in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [20, 22]:
    n = (1 << lg) - 6
    B = 16
    g = torch.Generator().manual_seed(lg)
    k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8, generator=g); k[:, 31] &= 0x1f
    table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
    d = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8, device=dev); d[:, :, 31] &= 0x1f
    out = torch.zeros((2, B, 96), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    res = {}
    for m, mode in enumerate(("ordinary",) if ORD_ONLY else ("fixed-base",) if FIX_ONLY else ("ordinary", "fixed-base")):
        tp = 0.0
        if mode == "fixed-base":
            t0 = time.perf_counter(); eng.bases_precompute(table, 0); tp = time.perf_counter() - t0
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out[m].data_ptr()); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out[m].data_ptr())
        eng.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 3 * 1e3, tp * 1e3)
    if ORD_ONLY or FIX_ONLY:
        mode = "ordinary" if ORD_ONLY else "fixed-base"
        print("2^%d - 6 points x %d: %s %.2f ms" % (lg, B, mode, res[mode][0]), flush=True)
        eng.bases_free(table)
        continue
    same = eng.g1_batch_to_affine(bytes(out[0].cpu().numpy().tobytes())) == eng.g1_batch_to_affine(bytes(out[1].cpu().numpy().tobytes()))
    print("2^%d - 6 points x %d: ordinary %.2f ms | fixed-base levels %.2f ms (precompute %.0f ms) | results equal: %s"
          % (lg, B, res["ordinary"][0], res["fixed-base"][0], res["fixed-base"][1], same), flush=True)
    eng.bases_free(table)
