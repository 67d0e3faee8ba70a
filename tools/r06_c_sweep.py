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
import numpy as np
for ndig in (13, 12, 11):
    d = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8, device=dev)
    if ndig == 13:
        d[:, :, 31] &= 0x1f          # uniform below 2^253 < r: the headline workload
    else:
        # ndig non-zero digits, none of them above 2^19 (bit 19 of every 20-bit digit clear): no carries, so the dropped top
        # digits stay empty instead of collecting a carry from half the scalars (which lands 2^21 keys on sixteen slots —
        # measured beside this sweep: 172 ms per batch)
        mask = np.zeros(32, dtype=np.uint8)
        for bit in range(20 * ndig):
            if bit % 20 != 19:
                mask[bit // 8] |= 1 << (bit % 8)
        d &= torch.from_numpy(mask).to(dev)
    bits = ndig
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
    res[ndig] = (min(ts), stages)
    print("%2d digits per scalar: batch of %d %.2f ms (%.3f per MSM; runs %s) | per MSM, one stage bracketed at a time: %s" % (
        ndig, B, min(ts), min(ts) / B, " ".join("%.1f" % t for t in ts),
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

# small scalars (what instance columns of real circuits hold): few distinct top digits, so a handful of bucket slots take
# most of the keys — levels + their sort against the ordinary path, one batch each
eng.msm_set_tail_overlap(2)
plain = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)      # the same table without levels
for nbits in (64, 128, 253):
    d = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8, device=dev)
    if nbits < 253:
        d[:, :, nbits // 8:] = 0
    else:
        d[:, :, 31] &= 0x1f
    torch.cuda.synchronize()
    line = "%3d-bit scalars:" % nbits
    for mode in ("levels", "ordinary"):
        tb = table if mode == "levels" else plain
        eng.g1_msm_device_batch_async(tb, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.g1_msm_device_batch_async(tb, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        line += "  %s %.1f ms per batch of %d" % (mode, min(ts), B)
        stg = {}
        for st_i in range(len(names)):
            eng.profile_enable(True, only_stage=st_i); eng.profile_reset()
            eng.g1_msm_device_batch_async(tb, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
            v = eng.profile_stages()[names[st_i]]
            if v[1] and v[0] / v[1] > 0.03:
                stg[names[st_i].replace("msm_", "")] = v[0] / v[1]
            eng.profile_enable(False)
        line += " (" + " ".join("%s=%.2f" % kv for kv in stg.items()) + ")"
    print(line, flush=True)
    del d
