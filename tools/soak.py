#!/usr/bin/env python3
"""Soak: many asynchronous MSMs, synchronous MSMs of changing sizes and schema evaluations in one process; device memory and
host RSS must stay flat after the warm-up, every result must keep matching the first one.  python tools/soak.py [seconds]"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
from bench import gen_scalars
pkg = entry.load_package(); eng = pkg.H2Agg(0)
import importlib
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
agg = importlib.import_module(entry.PKG_NAME + ".aggregate")
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
n = 1 << 20
_, k = gen_scalars(1, n); _, s = gen_scalars(2, n)
dk = torch.from_numpy(k.copy()).to(dev); ds = torch.from_numpy(s.copy()).to(dev)
table = eng.bases_generate(dk.data_ptr(), n)
out = torch.zeros(96 * 8, dtype=torch.uint8, device=dev)
eng.msm_set_tail_overlap(2)
backend = agg.GpuBackend(pkg, eng)
pool = syn.point_pool(eng, 0xA66)
specs, lam = syn.make_proofs(pool, 4, 300)

# the from-bytes pipeline over SIX call shapes (the context keeps four recordings: least-recently-used ones are evicted and
# recorded again) and the chained host-buffer MSM
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
comp = eng.g1_batch_compress(b"".join(pool))
pool_c = [comp[32 * i:32 * i + 32] for i in range(len(pool))]
_, gk = gen_scalars(7, 1 << 12)
g_table = eng.bases_generate(torch.from_numpy(gk.copy()).to(dev).data_ptr(), 1 << 12)
shape = syn.CircuitShape(12, 60, pool)
vk = ver.VerifyingKey(eng, ver.encode_vk(shape, lambda p: p))
fr = syn.fr_stream(0xF00D)
vproofs = [([b"".join(fr() for _ in range(16))], shape.random_transcript(pool_c, 500 + i)) for i in range(6)]
h_bases = eng.bases_download(table, 0, 1 << 18)
h_scal = bytes(s[:1 << 18].tobytes())

def one_round():
    res = []
    for k in range(1, 7):
        res.append(ver.verify_aggregation(eng, [(vk, "soak", g_table, vproofs[:k])])[:3])
    res.append(eng.g1_batch_to_affine(eng.g1_msm(h_bases, h_scal)))
    for i in range(8):
        eng.g1_msm_device_async(table, ds.data_ptr(), n, out.data_ptr() + 96 * i)
    eng.synchronize()
    res.append(eng.g1_batch_to_affine(bytes(out.cpu().numpy())))   # Jacobian results are not canonical (addition order): compare affine
    for m in (1 << 18, 1000, (1 << 16) + 3, 7):
        res.append(eng.g1_batch_to_affine(eng.g1_msm_device(table, ds.data_ptr(), m)))
    def build(b, idx):
        return [syn.build_proof(b, mo.MultiOpenProof, specs[i])[0] for i in idx]
    res.append(b"".join(agg.aggregate_sharded(backend, build, 4, lam)))
    return res

def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0

first = one_round()
for _ in range(5):
    assert one_round() == first
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info(dev)[0]; rss0 = rss_mb()
t_end = time.time() + secs
rounds = 0
while time.time() < t_end:
    assert one_round() == first, "result changed in round %d" % rounds
    rounds += 1
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info(dev)[0]; rss1 = rss_mb()
print("soak ok: %d rounds (8 async 2^20-point MSMs + 4 MSMs of other sizes + one 4-proof aggregation + six from-bytes aggregations of "
      "different shapes + one chained host-buffer MSM each); device memory %+.1f MiB, host max RSS %+.1f MiB; recorded aggregations "
      "(hits, misses, kept) %s" % (rounds, (free0 - free1) / 2**20, rss1 - rss0, eng.verify_plan_stats()))
