"""three h2agg_g1_msm calls from page-locked host buffers (for a rocprofv3 --kernel-trace --memory-copy-trace run)
    python tools/pcie_one.py [log2n]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
from bench import gen_scalars
pkg = entry.load_package()
eng = pkg.H2Agg(0)
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
_, k_np = gen_scalars(1, n)
_, s_np = gen_scalars(2, n)
d_k = torch.from_numpy(k_np.copy()).cuda()
table = eng.bases_generate(d_k.data_ptr(), n)
bases = eng.bases_download(table, 0, n)
pb, ps = eng.host_alloc(64 * n), eng.host_alloc(32 * n)
ctypes.memmove(pb, bases, 64 * n)
ctypes.memmove(ps, bytes(s_np.tobytes()), 32 * n)
for _ in range(4):
    t0 = time.perf_counter()
    eng.g1_msm(pb, ps, n)
    print("%.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(0.01)
