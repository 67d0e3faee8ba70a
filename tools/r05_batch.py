#!/usr/bin/env python3
"""a batch of B MSMs over one 2^lg-point table the way the aggregation's instance columns run (one by one, tails overlapped and
deferred): ordinary path against fixed-base levels on ONE box, whole batch timed, then single stages bracketed one at a time
(one event pair per MSM does not switch the deferred tails off).   python tools/r05_batch.py [lg] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = (1 << lg) - 6
g = torch.Generator().manual_seed(lg)
k = torch.randint(0, 256, (1 << lg, 32), dtype=torch.uint8, generator=g); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << lg)
d = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8, device=dev); d[:, :, 31] &= 0x1f
out = torch.zeros((2, B, 96), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
names = list(eng.profile_stages().keys())
for m, mode in enumerate(("ordinary", "fixed-base")):
    if mode == "fixed-base":
        eng.bases_precompute(table, 0)
    def batch():
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out[m].data_ptr())
    batch(); eng.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); batch(); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    line = "%-10s %6.2f ms per batch of %d (%.3f per MSM; runs %s) |" % (mode, min(ts), B, min(ts) / B, " ".join("%.1f" % t for t in ts))
    for st in (1, 2, 4, 5):
        eng.profile_enable(True, only_stage=st); eng.profile_reset()
        batch(); eng.synchronize()
        v = eng.profile_stages()[names[st]]
        line += " %s=%.3f" % (names[st].replace("msm_", ""), v[0] / max(1, v[1]))
        eng.profile_enable(False)
    print(line, flush=True)
same = eng.g1_batch_to_affine(bytes(out[0].cpu().numpy().tobytes())) == eng.g1_batch_to_affine(bytes(out[1].cpu().numpy().tobytes()))
print("results equal:", same)
