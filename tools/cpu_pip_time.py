"""per-thread speed of the oracle's CPU Pippenger on this host: n points, c, threads"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cref, bn254 as O
import numpy as np
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
rng = np.random.default_rng(5)
ks = rng.integers(0, 256, (n, 32), dtype=np.uint8); ks[:, 31] &= 0x1f
G = O.aff_to_bytes(O.G1)
t0 = time.perf_counter()
bases = cref.g1_batch_to_affine(cref.g1_batch_scalar_mul(G * 4096, ks[:4096].tobytes(), 4096), 4096) * (n // 4096)
print("bases %.2f s" % (time.perf_counter() - t0))
sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[:, 31] &= 0x1f
for c, thr, jpt in ((13, 1, 1), (15, 1, 1), (16, 1, 1), (13, 16, 4), (15, 16, 4), (16, 16, 4), (16, 8, 4), (16, 4, 4)):
    t0 = time.perf_counter()
    cref.msm_pippenger2(bases, sc.tobytes(), n, c, thr, jpt)
    dt = time.perf_counter() - t0
    W = (255 + c - 1) // c
    print("n=2^%d c=%d threads=%d jpt=%d: %.3f s  %.2f M points/s  %.0f ns per insertion per thread" % (
        int(sys.argv[1]) if len(sys.argv) > 1 else 16, c, thr, jpt, dt, n / dt / 1e6, dt * thr / (n * W) * 1e9))
