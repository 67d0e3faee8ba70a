"""where does the occasional 20-30 ms stall of a 64-proof host-backend call sit after device-backend calls?  (phase trace)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["H2AGG_TRACE_PHASES"] = "1"
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
syn = importlib.import_module(entry.PKG_NAME + ".synthetic"); ver = importlib.import_module(entry.PKG_NAME + ".verifier")
from bench import gen_scalars
dev = torch.device('cuda', 0)
_, gk = gen_scalars(7, 1 << 17)
g_table = eng.bases_generate(torch.from_numpy(gk.copy()).to(dev).data_ptr(), 1 << 17); eng.bases_precompute(g_table)
pool = syn.point_pool(eng, 0xA66); comp = eng.g1_batch_compress(b"".join(pool)); pool_c = [comp[32 * i:32 * i + 32] for i in range(len(pool))]
shape = syn.CircuitShape(17, 300, pool); vk = ver.VerifyingKey(eng, ver.encode_vk(shape, lambda p: p)); fr = syn.fr_stream(0xF00D)
n = 64
proofs = [([b"".join(fr() for _ in range(64))], shape.random_transcript(pool_c, 100 + i)) for i in range(n)]
arg = [(vk, "syn", g_table, proofs)]
for be, reps in (("host", 6), ("device", 3), ("host", 12)):
    eng.transcript_configure(be)
    for r in range(reps):
        t0 = time.perf_counter(); ver.verify_aggregation(eng, arg); dt = time.perf_counter() - t0
        print("%s rep %d: %.2f ms" % (be, r, dt * 1e3), file=sys.stderr, flush=True)
