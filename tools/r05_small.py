import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8); k[:, 31] &= 0x1f
table = eng.bases_generate(k.to(dev).data_ptr(), n)
d = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev); d[:, 31] &= 0x1f
a = eng.g1_batch_to_affine(eng.g1_msm_device(table, d.data_ptr(), n))
print("ordinary ok", flush=True)
eng.bases_precompute(table, 20)
print("precompute ok", flush=True)
b = eng.g1_batch_to_affine(eng.g1_msm_device(table, d.data_ptr(), n))
print("equal", a == b, flush=True)
