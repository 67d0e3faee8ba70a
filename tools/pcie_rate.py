#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point h2agg_g1_msm (bases + scalars cross PCIe every call,
bases are converted to Montgomery form on the device).  This is NOT bench.py's `value` (inputs resident)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as entry
from bench import gen_scalars

pkg = entry.load_package()
extra = [a for a in sys.argv if a.startswith('--extra-streams=')]
_keep = [torch.cuda.Stream() for _ in range(int(extra[0].split('=')[1]) if extra else 0)]   # (HIP maps streams onto few hardware queues)
for st in _keep:
    with torch.cuda.stream(st):
        torch.zeros(8, device='cuda').add_(1)
eng = pkg.H2Agg(0)
log2n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
if '--torch-stream' in sys.argv:   # what bench.py does: the library works on torch's current stream
    _st = torch.cuda.Stream(); torch.cuda.set_stream(_st)   # (the default stream's handle is 0 = "the context's own stream")
    eng.set_stream(_st.cuda_stream)
n = 1 << log2n
_, k_np = gen_scalars(1, n)
_, s_np = gen_scalars(2, n)
d_k = torch.from_numpy(k_np.copy()).cuda()
table = eng.bases_generate(d_k.data_ptr(), n)
bases = eng.bases_download(table, 0, n)
scalars = bytes(s_np.tobytes())
eng.g1_msm(bases, scalars)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    out = eng.g1_msm(bases, scalars)
dt = (time.perf_counter() - t0) / reps
# the same call from page-locked buffers (h2agg_host_alloc)
import ctypes
pb, ps = eng.host_alloc(64 * n), eng.host_alloc(32 * n)
ctypes.memmove(pb, bases, 64 * n)
ctypes.memmove(ps, scalars, 32 * n)
eng.g1_msm(pb, ps, n)
t0 = time.perf_counter()
for _ in range(reps):
    out3 = eng.g1_msm(pb, ps, n)
dt3 = (time.perf_counter() - t0) / reps
assert eng.g1_batch_to_affine(out3) == eng.g1_batch_to_affine(out)
eng.host_free(pb)
eng.host_free(ps)
d_s = torch.from_numpy(s_np.copy()).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out2 = eng.g1_msm_device(table, d_s.data_ptr(), n)
dt2 = (time.perf_counter() - t0) / reps
assert eng.g1_batch_to_affine(out) == eng.g1_batch_to_affine(out2)
print("n=2^%d  from page-locked buffers: %.2f ms/call = %.1f Mpoints/s" % (log2n, dt3 * 1e3, n / dt3 / 1e6))
print("n=2^%d  host-buffer h2agg_g1_msm: %.2f ms/call = %.1f Mpoints/s (moves %d MiB over PCIe per call);  "
      "resident h2agg_g1_msm_device (synchronous, single MSM latency): %.2f ms = %.1f Mpoints/s"
      % (log2n, dt * 1e3, n / dt / 1e6, 96 * n >> 20, dt2 * 1e3, n / dt2 / 1e6))
