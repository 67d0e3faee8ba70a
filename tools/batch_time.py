import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
eng.msm_set_tail_overlap(2)
dev = torch.device("cuda:0")
n = (1 << 17) - 6
k = torch.randint(0, 256, (1 << 17, 32), dtype=torch.uint8); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), 1 << 17)
for B in (1, 2, 4, 8, 16):
    s = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8); s[:, :, 31] &= 0x1f
    d = s.to(dev); out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
    eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr())
        eng.synchronize()
    dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        for q in range(B):
            eng.g1_msm_device_async(table, d[q].data_ptr(), n, out[q].data_ptr())
        eng.synchronize()
    dt2 = (time.perf_counter() - t0) / 5
    print("batch %2d x 2^17: batched %.3f ms (%.0f M points/s)   one by one %.3f ms" % (B, dt * 1e3, B * n / dt / 1e6, dt2 * 1e3))
for c in (0, 13, 14, 15, 16):
    eng.msm_configure(window_bits=c)
    B = 16
    s = torch.randint(0, 256, (B, n, 32), dtype=torch.uint8); s[:, :, 31] &= 0x1f
    d = s.to(dev); out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
    eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr()); eng.synchronize()
    eng.profile_reset(); eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.g1_msm_device_batch_async(table, d.data_ptr(), n, B, out.data_ptr())
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 5
    eng.profile_enable(False)
    st = eng.profile_stages()
    print("c=%2d batch 16: %.3f ms  " % (c, dt * 1e3) + " ".join("%s=%.2f" % (k.replace("msm_", ""), v[0] / 5) for k, v in st.items()))
