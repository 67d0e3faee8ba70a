"""timing of adversarial scalar distributions (over-long buckets): all scalars equal / half equal / 1 % equal, 2^20 points.
    python tools/skew_time.py [log2n]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
eng = pkg.H2Agg(0)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
from bench import gen_scalars
_, k = gen_scalars(1, n)
_, s = gen_scalars(2, n)
dev = torch.device('cuda', 0)
dk = torch.from_numpy(k.copy()).to(dev)
t = eng.bases_generate(dk.data_ptr(), n)
out = torch.zeros(96 * 8, dtype=torch.uint8, device=dev)
for name, frac in (("random", 0.0), ("1% equal", 0.01), ("half equal", 0.5), ("all equal", 1.0)):
    sc = s.copy().reshape(n, -1)
    m = int(n * frac)
    if m:
        sc[:m] = sc[0]
    ds = torch.from_numpy(sc.reshape(-1).copy()).to(dev)
    for glv in (0, 1):
        eng.msm_configure_glv(1 if glv else -1)
        for i in range(2):
            eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr())
        eng.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr() + 96 * i)
        eng.synchronize()
        print("%-11s glv=%d  %.3f ms/MSM" % (name, glv, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
