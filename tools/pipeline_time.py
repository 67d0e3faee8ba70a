"""h2agg_verify_aggregation end to end (proof bytes in, pairing verdict out) at several batch sizes, on both sponge backends.
    python tools/pipeline_time.py [proofs ...]      (default 4 16 64)
Same synthetic P = 347 key as bench.py's `aggregate.full_pipeline` leg; prints ms per aggregation and proofs/s, and checks
that the two backends return the same pair and lambda."""
import importlib, sys, time, types
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
eng = pkg.H2Agg(0)
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
from bench import gen_scalars
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4, 16, 64]
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
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
import gc
gc.disable()   # (the 35-45 ms calls this tool used to report as `max` were CPython's full collections — tools/stall_hunt.py — not the library)
print("host worker threads:", pkg.host_threads())
for n in sizes:
    arg = [(vk, "syn", g_table, proofs_all[:n])]
    res = {}
    for be in (("device", "host", "auto", "auto/recorded-every-call") if "--host-first" not in sys.argv else ("host", "auto", "device", "host")):
        eng.transcript_configure(be.split("/")[0])
        # (the host-side recording of a call is kept and reused by later calls of the same shape; the last row switches that off)
        eng.debug_configure("plan_cache", 0 if "/" in be else 1)
        for pair in (True, False):
            a = ver.verify_aggregation(eng, arg, g2 if pair else None, g2 if pair else None)
            reps = 9
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                b = ver.verify_aggregation(eng, arg, g2 if pair else None, g2 if pair else None)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            dt = ts[len(ts) // 2]          # median; min and max beside it (a box hiccup of 20 ms is not the rate)
            assert a[:3] == b[:3]
            res[be] = a[:3]
            print("%3d proofs  %-24s  %s  %8.3f ms  %8.1f proofs/s   (min %.3f max %.3f)" % (
                n, be, "with pairing" if pair else "no pairing  ", dt * 1e3, n / dt, ts[0] * 1e3, ts[-1] * 1e3), flush=True)
    assert res["device"] == res["host"] == res["auto"] == res.get("auto/recorded-every-call", res["auto"]), "backends disagree"
eng.debug_configure("plan_cache", 1)
eng.transcript_configure("auto")
print("recorded aggregations: hits %d, misses %d, kept %d" % eng.verify_plan_stats())
vk.close()
