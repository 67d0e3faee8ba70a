"""one adversarial distribution under a profiler: python tools/skew_one.py <frac equal> <glv -1|1> [log2n]"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
eng = pkg.H2Agg(0)
frac = float(sys.argv[1]); glv = int(sys.argv[2])
n = 1 << int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
from bench import gen_scalars
_, k = gen_scalars(1, n)
_, s = gen_scalars(2, n)
dev = torch.device('cuda', 0)
dk = torch.from_numpy(k.copy()).to(dev)
t = eng.bases_generate(dk.data_ptr(), n)
out = torch.zeros(96 * 8, dtype=torch.uint8, device=dev)
sc = s.copy().reshape(n, -1)
m = int(n * frac)
if m:
    sc[:m] = sc[0]
ds = torch.from_numpy(sc.reshape(-1).copy()).to(dev)
eng.msm_configure_glv(glv)
for i in range(4):
    eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr())
    eng.synchronize()
