#!/usr/bin/env python3
"""timing experiment (timing build only, WRONG results): the fixed-base accumulation with its gathers confined to the first
2^k table entries — how much of the 3.9 ms is the gathers' way to HBM.   python tools/r05_mask.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = 22; n = (1 << lg) - 6
g = torch.Generator().manual_seed(lg)
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8, generator=g); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
d = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev); d[:, 31] &= 0x1f
eng.bases_precompute(table, 0)
o1 = torch.zeros(96, dtype=torch.uint8, device=dev)
for bits in (32, 26, 24, 22, 20, 18):
    eng.debug_configure("fb_mask", (1 << bits) - 1 if bits < 32 else -1)
    eng.g1_msm_device_async(table, d.data_ptr(), n, o1.data_ptr()); eng.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(3):
        eng.g1_msm_device_async(table, d.data_ptr(), n, o1.data_ptr())
    eng.synchronize()
    st = eng.profile_stages()
    print("mask 2^%d entries (%d MiB): accumulate %.3f ms" % (bits, (64 << min(bits, 26)) >> 20, st["msm_accumulate"][0] / 3), flush=True)
    eng.profile_enable(False)
