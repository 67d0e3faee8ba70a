#!/usr/bin/env python3
"""accumulate stage alone (single MSMs, no overlap) at 2^22: ordinary c = 17 / 16, fixed-base c = 20 with 1 / 2 lanes per bucket"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22; n = (1 << lg) - 6
g = torch.Generator().manual_seed(lg)
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8, generator=g); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
d = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev); d[:, 31] &= 0x1f
o1 = torch.zeros(96, dtype=torch.uint8, device=dev)
eng.msm_set_tail_overlap(0)
def run(tag):
    eng.g1_msm_device_async(table, d.data_ptr(), n, o1.data_ptr()); eng.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(4):
        eng.g1_msm_device_async(table, d.data_ptr(), n, o1.data_ptr())
    eng.synchronize()
    st = eng.profile_stages()
    print("%-28s " % tag + "  ".join("%s=%.3f" % (a.replace("msm_", ""), v[0] / 4) for a, v in st.items() if v[1]), flush=True)
    eng.profile_enable(False)
eng.msm_configure_glv(-1)
run("ordinary auto (c=17)")
eng.msm_configure(16, 0, 0); run("ordinary c=16"); eng.msm_configure(0, 0, 0)
eng.bases_precompute(table, 0)
run("fixed-base c=20")
eng.msm_configure_lanes_per_bucket(2); run("fixed-base c=20 lpb=2"); eng.msm_configure_lanes_per_bucket(0)
eng.msm_configure(0, 0, 2000); run("fixed-base big=2000"); eng.msm_configure(0, 0, 0)
