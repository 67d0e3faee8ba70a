"""window widths above 16 at large sizes (the digit-major sort and the two-dimensional reduction only exist for c = 16; wider
windows take the packed sort and the segment reduction): python tools/big_window_time.py [log2n ...]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
eng = pkg.H2Agg(0)
from bench import gen_scalars
dev = torch.device('cuda', 0)
for lg in [int(a) for a in sys.argv[1:]] or [22, 24]:
    n = 1 << lg
    _, k = gen_scalars(1, n)
    _, s = gen_scalars(2, n)
    dk = torch.from_numpy(k.copy()).to(dev); ds = torch.from_numpy(s.copy()).to(dev)
    t = eng.bases_generate(dk.data_ptr(), n)
    out = torch.zeros(96 * 8, dtype=torch.uint8, device=dev)
    K = max(3, min(20, (1 << 25) // n))
    ref = None
    for glv in (-1, 1):
        for c in (16, 17, 18, 19, 20):
            eng.msm_configure(window_bits=c)
            eng.msm_configure_glv(glv)
            try:
                for i in range(2):
                    eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr())
                eng.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr() + 96 * (i % 8))
                eng.synchronize()
                dt = (time.perf_counter() - t0) / K
                got = bytes(out[:96].cpu().numpy())
                aff = eng.g1_batch_to_affine(got)
                ref = ref or aff
                print("2^%d glv=%2d c=%2d  %8.3f ms  %6.1f Mpts/s  %s" % (lg, glv, c, dt * 1e3, n / dt / 1e6, "ok" if aff == ref else "MISMATCH"), flush=True)
            except Exception as ex:
                print("2^%d glv=%2d c=%2d  %s" % (lg, glv, c, str(ex)[:80]), flush=True)
    eng.msm_configure(window_bits=0); eng.msm_configure_glv(0)
    eng.bases_free(t)
