import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as entry
from bench import gen_scalars, R_MOD
pkg = entry.load_package(); eng = pkg.H2Agg(0)
g_aff = (1).to_bytes(32,"little") + (2).to_bytes(32,"little")
n = 1 << 22
seed = 0x48324147
ks, k_np = gen_scalars(seed, n); ss, s_np = gen_scalars(seed + 1, n)
d_k = torch.from_numpy(k_np.copy()).cuda(); d_s = torch.from_numpy(s_np.copy()).cuda()
table = eng.bases_generate(d_k.data_ptr(), n)
total = sum(k*s for k, s in zip(ks, ss)) % R_MOD
want = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(g_aff, total.to_bytes(32,"little")))
for name, cfg in (("default", (0,0)), ("direct", (0,-1)), ("sub10", (10,0))):
    eng.msm_configure_sort(*cfg)
    got = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
    print(name, got == want, flush=True)
eng.msm_configure_sort()
# digits of the last scalar
s = ss[-1]; c = 16; carry = 0
for w in range(16):
    raw = ((s >> (16*w)) & 0xffff) + carry
    neg = raw > 0x8000; carry = 1 if neg else 0
    mag = (0x10000 - raw) if neg else raw
    print(w, mag - 1, (mag - 1) & 511, neg)
# check the base table against k*G for a few indices incl. the last
idx = [0, 1, n // 2, n - 2, n - 1]
for i in idx:
    b = eng.bases_download(table, i, 1)
    w_ = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(g_aff, ks[i].to_bytes(32, "little")))
    print("base", i, b == w_)
