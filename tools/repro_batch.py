"""reproducer attempt: small fixed-base batches in overlap mode, then the first larger batch on a fresh context"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.Generator(np.random.PCG64(5))
k = rng.integers(0, 256, size=(4096, 32), dtype=np.uint8); k[:, 31] &= 0x1f
for rep in range(reps):
    eng = pkg.H2Agg(0)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # (handle 0 = the context's own stream: torch's fills below are NOT ordered before the library — on purpose, this is the reproducer)
    d_k = torch.from_numpy(k.copy()).to(dev)
    t = eng.bases_generate(d_k.data_ptr(), 4096)
    eng.bases_precompute(t)
    eng.msm_set_tail_overlap(2)
    n = 4090
    outs = {}
    for batch in (2, 2, 2, 8, 2, 16, 8):
        d_s = torch.randint(0, 256, (batch, n, 32), dtype=torch.uint8, device=dev); d_s[:, :, 31] &= 0x1f
        d_o = torch.zeros((batch, 96), dtype=torch.uint8, device=dev)
        eng.g1_msm_device_batch_async(t, d_s.data_ptr(), n, batch, d_o.data_ptr())
        aff = eng.g1_batch_to_affine_device(d_o.data_ptr(), batch)
        # the same through single MSMs
        one = eng.g1_batch_to_affine(eng.g1_msm_device(t, d_s[batch - 1].data_ptr(), n))
        assert aff[64 * (batch - 1):] == one, (rep, batch)
        # a couple of small ordinary MSMs in between (what an evaluation issues)
        for m in (614, 5):
            eng.g1_msm_device_async(t, d_s[0].data_ptr(), m, d_o[0].data_ptr())
        eng.synchronize()
    eng.bases_free(t)
    eng.close()
print("repro loop ok:", reps)
