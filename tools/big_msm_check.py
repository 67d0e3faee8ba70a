import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
rng = np.random.Generator(np.random.PCG64(9))
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 31] &= 0x1f
s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); s[:, 31] &= 0x1f
t0 = time.time()
# sum k_i * s_i mod r with 64-bit limb arithmetic via Python ints in chunks
kb, sb = k.tobytes(), s.tobytes()
tot = 0
for i in range(n):
    tot += int.from_bytes(kb[32*i:32*i+32], "little") * int.from_bytes(sb[32*i:32*i+32], "little")
tot %= R
print("host sum %.1f s" % (time.time() - t0), flush=True)
d_k = torch.from_numpy(k).to(dev); d_s = torch.from_numpy(s).to(dev)
table = eng.bases_generate(d_k.data_ptr(), n)
g = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
want = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(g, tot.to_bytes(32, "little")))
for glv in (-1, 1):
    eng.msm_configure_glv(glv)
    t0 = time.perf_counter()
    got = eng.g1_batch_to_affine(eng.g1_msm_device(table, d_s.data_ptr(), n))
    dt = time.perf_counter() - t0
    print("2^%d glv=%d: %s  %.1f ms (first call)" % (lg, glv, "OK" if got == want else "MISMATCH", dt * 1e3), flush=True)
    t0 = time.perf_counter(); eng.g1_msm_device(table, d_s.data_ptr(), n); print("   second call %.1f ms = %.0f M points/s" % ((time.perf_counter()-t0)*1e3, n/(time.perf_counter()-t0)/1e6))
