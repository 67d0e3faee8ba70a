#!/usr/bin/env python3
"""would wider windows (12 / 11 insertions per scalar instead of 13) bring a batch of 16 MSMs over a
2^22-point table with fixed-base levels under 66 ms?  Measured pieces, one box:
  * the batch (h2agg_g1_msm_device_batch_async, 16 x (2^22 - 6) scalars, c = 20 levels) with scalars of 254 / 240 / 220 bits:
    13 / 12 / 11 non-zero digits per scalar, i.e. the accumulation and the sort at the insertion counts of c = 20 / 22 / 24
    (same kernels, same bucket space: what a wider window saves in insertions, and nothing of what it costs);
  * the bucket reduction of ONE MSM (the stages behind the accumulation, tails in-stream so that they are timed alone), which
    scales with the bucket count: x 4 at c = 22, x 16 at c = 24.
python tools/r06_c_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg, B = 22, 16
n = (1 << lg) - 6
g = torch.Generator().manual_seed(lg)
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8, generator=g); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
eng.bases_precompute(table, 0)
out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
names = None
res = {}
for bits in (254, 240, 220):
    d = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8, device=dev)
    full, rem = bits // 8, bits % 8
    if full < 32:
        d[:, :, full + (1 if rem else 0):] = 0
        if rem:
            d[:, :, full] &= (1 << rem) - 1
    if bits == 254:
        d[:, :, 31] &= 0x1f          # < 2^253 < r
    torch.cuda.synchronize()
    def batch():
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr())
    batch(); eng.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); batch(); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    names = names or list(eng.profile_stages().keys())
    stages = {}
    for st in range(len(names)):
        eng.profile_enable(True, only_stage=st); eng.profile_reset()
        batch(); eng.synchronize()
        v = eng.profile_stages()[names[st]]
        if v[1]:
            stages[names[st]] = v[0] / v[1]
        eng.profile_enable(False)
    res[bits] = (min(ts), stages)
    print("%3d-bit scalars (%2d digits): batch of %d %.2f ms (%.3f per MSM; runs %s) | per MSM, one stage bracketed at a time: %s" % (
        bits, (bits + 19) // 20, B, min(ts), min(ts) / B, " ".join("%.1f" % t for t in ts),
        " ".join("%s=%.3f" % (a.replace("msm_", ""), b) for a, b in stages.items())), flush=True)
    del d
# one MSM alone with its tail in-stream: the reduction behind the accumulation, timed stage by stage
d1 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev); d1[:, 31] &= 0x1f
torch.cuda.synchronize()
eng.msm_set_tail_overlap(0)
o1 = torch.zeros(96, dtype=torch.uint8, device=dev)
eng.g1_msm_device_async(table, d1.data_ptr(), n, o1.data_ptr()); eng.synchronize()
eng.profile_enable(True); eng.profile_reset()
for _ in range(5):
    eng.g1_msm_device_async(table, d1.data_ptr(), n, o1.data_ptr())
eng.synchronize()
st = eng.profile_stages()
eng.profile_enable(False)
print("one MSM, tails in-stream, every stage bracketed (ms per MSM):", " ".join("%s=%.3f" % (a.replace("msm_", ""), v[0] / v[1]) for a, v in st.items() if v[1]))
