"""memory safety with non-canonical scalars: random 256-bit values (and all-ones) through every MSM route must end in
ERR_NONCANONICAL, never in a fault"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
eng = pkg.H2Agg(0)
rng = np.random.Generator(np.random.PCG64(5))
k = rng.integers(0, 256, size=(1 << 17, 32), dtype=np.uint8); k[:, 31] &= 0x1f
d_k = torch.from_numpy(k.copy()).to(dev)
t_small = eng.bases_generate(d_k.data_ptr(), 4096)
eng.bases_precompute(t_small)
t_big = eng.bases_generate(d_k.data_ptr(), 1 << 17)
torch.cuda.synchronize()
def attempt(what, fn):
    try:
        fn()
        eng.synchronize()
        print(what, "-> no error reported")
    except pkg.H2AggError as e:
        print(what, "->", e.code)
for fill in ("random", "ones"):
    for ovl in (0, 2):
        eng.msm_set_tail_overlap(ovl)
        for batch in (2, 8):
            n = 4090
            g = torch.randint(0, 256, (batch, n, 32), dtype=torch.uint8, device=dev) if fill == "random" else torch.full((batch, n, 32), 255, dtype=torch.uint8, device=dev)
            o = torch.zeros((batch, 96), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            attempt("%s fixed-base batch %d overlap %d" % (fill, batch, ovl), lambda: eng.g1_msm_device_batch_async(t_small, g.data_ptr(), n, batch, o.data_ptr()))
        for n in (5, 614, 4090, 1 << 16, 1 << 17):
            g = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev) if fill == "random" else torch.full((n, 32), 255, dtype=torch.uint8, device=dev)
            o = torch.zeros(96, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for glv in (1, -1):
                eng.msm_configure_glv(glv)
                attempt("%s plain n=%d glv %d overlap %d" % (fill, n, glv, ovl), lambda: eng.g1_msm_device_async(t_big, g.data_ptr(), n, o.data_ptr()))
            eng.msm_configure_glv(0)
print("garbage scalars: survived")
