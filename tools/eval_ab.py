"""A/B of the evaluation's launch structure inside h2agg_verify_aggregation (same process, alternating):
    python tools/eval_ab.py [proofs ...]      (default 4 16)
keys: eval_split (both multi_exps as one split MSM) and small_sort (one-launch sort); prints the median ms per aggregation."""
import importlib, sys, time, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
eng = pkg.H2Agg(0)
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
from bench import gen_scalars
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4, 16]
dev = torch.device('cuda', 0)
_, gk = gen_scalars(7, 1 << 17)
g_table = eng.bases_generate(torch.from_numpy(gk.copy()).to(dev).data_ptr(), 1 << 17)
eng.bases_precompute(g_table)
pool = syn.point_pool(eng, 0xA66)
comp = eng.g1_batch_compress(b"".join(pool))
pool_c = [comp[32 * i:32 * i + 32] for i in range(len(pool))]
shape = syn.CircuitShape(17, 300, pool)
vk = ver.VerifyingKey(eng, ver.encode_vk(shape, lambda p: p))
fr = syn.fr_stream(0xF00D)
proofs_all = [([b"".join(fr() for _ in range(64))], shape.random_transcript(pool_c, 100 + i)) for i in range(max(sizes))]
gc.disable()
eng.transcript_configure("auto")
for n in sizes:
    arg = [(vk, "syn", g_table, proofs_all[:n])]
    want = None
    for rnd in range(3):
        for split, small in ((1, 1), (0, 1), (0, 0)):
            eng.debug_configure("eval_split", split)
            eng.debug_configure("small_sort", small)
            a = ver.verify_aggregation(eng, arg, None, None)
            want = want or a[:3]
            assert a[:3] == want
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                ver.verify_aggregation(eng, arg, None, None)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            print("%3d proofs  eval_split=%d small_sort=%d   %7.3f ms (min %.3f)" % (n, split, small, ts[len(ts) // 2] * 1e3, ts[0] * 1e3), flush=True)
eng.debug_configure("eval_split", 1)
eng.debug_configure("small_sort", 1)
vk.close()
