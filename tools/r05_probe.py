#!/usr/bin/env python3
"""round-5 probe: where a batch of MSMs over one 2^22-point table spends its time — ordinary path against fixed-base levels
through the two-array sort (limit lifted with the debug key) — per stage.   python tools/r05_probe.py [lg] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = (1 << lg) - 6
g = torch.Generator().manual_seed(lg)
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8, generator=g); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
d = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8, device=dev); d[:, :, 31] &= 0x1f
out = torch.zeros((2, B, 96), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for m, mode in enumerate(("ordinary", "fixed-base")):
    if mode == "fixed-base":
        t0 = time.perf_counter(); eng.bases_precompute(table, int(os.environ.get("PRE_C", "0"))); print("precompute %.0f ms" % ((time.perf_counter() - t0) * 1e3))
    eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out[m].data_ptr()); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out[m].data_ptr())
    eng.synchronize()
    print("%s: %.2f ms per batch of %d = %.2f ms per MSM" % (mode, (time.perf_counter() - t0) / 2 * 1e3, B, (time.perf_counter() - t0) / 2 / B * 1e3), flush=True)
    eng.profile_enable(True); eng.profile_reset()
    eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out[m].data_ptr()); eng.synchronize()
    st = eng.profile_stages()
    print("   stages (ms per MSM): " + "  ".join("%s=%.3f" % (a.replace("msm_", ""), v[0] / B) for a, v in st.items() if v[1]), flush=True)
    eng.profile_enable(False)
same = eng.g1_batch_to_affine(bytes(out[0].cpu().numpy().tobytes())) == eng.g1_batch_to_affine(bytes(out[1].cpu().numpy().tobytes()))
print("results equal:", same)
# single MSMs, no overlap: the stages alone (fixed-base levels are set by now); entries of a bucket in level order / scrambled
for scr in (0, 1):
    eng.debug_configure("pre_big", scr)
    eng.msm_set_tail_overlap(0)
    o1 = torch.zeros(96, dtype=torch.uint8, device=dev)
    eng.g1_msm_device_async(table, d[0].data_ptr(), n, o1.data_ptr()); eng.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(3):
        eng.g1_msm_device_async(table, d[0].data_ptr(), n, o1.data_ptr())
    eng.synchronize()
    st = eng.profile_stages()
    print("single, scramble=%d: " % scr + "  ".join("%s=%.3f" % (a.replace("msm_", ""), v[0] / 3) for a, v in st.items() if v[1]), flush=True)
    eng.profile_enable(False)
eng.debug_configure("pre_big", 0)
try:
    eng.debug_configure("fb_dump", 1)
except Exception as e:
    pass
