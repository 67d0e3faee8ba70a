#!/usr/bin/env python3
"""Stage times of the batched fixed-base instance-commitment MSM (B x (2^17 - 6) scalars against one precomputed table):
python tools/instance_batch_stages.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = 17
n = (1 << lg) - 6
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
eng.bases_precompute(table, 0)
for B in [int(a) for a in sys.argv[1:]] or [4, 16]:
    s = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8); s[:, :, 31] &= 0x1f
    d = s.to(dev); out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
    for ovl in (0, 2):
        eng.msm_set_tail_overlap(ovl)
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        dt = (time.perf_counter() - t0) / 5
        eng.profile_reset(); eng.profile_enable(True)
        for _ in range(3):
            eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        eng.profile_enable(False)
        st = eng.profile_stages()
        print("B=%d overlap=%d  %.3f ms per batch | " % (B, ovl, dt * 1e3) + " ".join("%s=%.3f" % (k.replace("msm_", ""), v[0] / 3) for k, v in st.items() if v[1]), flush=True)
