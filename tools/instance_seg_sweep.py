"""reduce-segment sweep of the batched fixed-base instance MSM (B x (2^17 - 6) scalars): python tools/instance_seg_sweep.py"""
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
eng.msm_set_tail_overlap(2)
for B in (1, 2, 4, 8, 16, 64):
    s = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8); s[:, :, 31] &= 0x1f
    d = s.to(dev); out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
    ref = None
    for seg in (8, 16, 32, 8, 16, 32):
        eng.msm_configure(0, seg, 0)
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        aff = eng.g1_batch_to_affine(bytes(out.cpu().numpy()))
        ref = ref or aff
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
            ts.append(time.perf_counter() - t0)
        print("B=%d seg=%3d  %.3f ms %s" % (B, seg, sorted(ts)[3] * 1e3, "ok" if aff == ref else "MISMATCH"), flush=True)
    eng.msm_configure(0, 0, 0)
