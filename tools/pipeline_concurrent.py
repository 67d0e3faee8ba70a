"""Aggregated proofs per second of ONE GPU when several host threads drive it, each with its own context: a call's host
phases (sponges on the worker pool, pairing) run while another call's device phases (instance MSMs, evaluation) do.
    python tools/pipeline_concurrent.py [proofs per call] [seconds]
Same synthetic P = 347 key as bench.py's full_pipeline leg; every thread alternates between two sets of proofs and checks
that each set keeps giving the same pair."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
from bench import gen_scalars
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
dev = torch.device("cuda", 0)


class Worker:
    def __init__(self, idx):
        self.eng = pkg.H2Agg(0)
        _, gk = gen_scalars(7, 1 << 17)
        self.g_table = self.eng.bases_generate(torch.from_numpy(gk.copy()).to(dev).data_ptr(), 1 << 17)
        self.eng.bases_precompute(self.g_table)
        pool = syn.point_pool(self.eng, 0xA66)
        comp = self.eng.g1_batch_compress(b"".join(pool))
        pool_c = [comp[32 * i:32 * i + 32] for i in range(len(pool))]
        shape = syn.CircuitShape(17, 300, pool)
        self.vk = ver.VerifyingKey(self.eng, ver.encode_vk(shape, lambda p: p))
        fr = syn.fr_stream(0xF00D + idx)
        proofs = [([b"".join(fr() for _ in range(64))], shape.random_transcript(pool_c, 1000 * idx + i)) for i in range(2 * k)]
        self.sets = [[(self.vk, "syn", self.g_table, proofs[:k])], [(self.vk, "syn", self.g_table, proofs[k:])]]
        self.first = [ver.verify_aggregation(self.eng, a, g2, g2)[:3] for a in self.sets]
        self.calls = 0

    def run(self, t_end):
        r = 0
        while time.perf_counter() < t_end:
            got = ver.verify_aggregation(self.eng, self.sets[r & 1], g2, g2)[:3]
            assert got == self.first[r & 1]
            r += 1
        self.calls = r


for nthreads in ([int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 2, 3, 4)):
    ws = [Worker(i) for i in range(nthreads)]
    t0 = time.perf_counter()
    ts = [threading.Thread(target=w.run, args=(t0 + secs,)) for w in ws]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    calls = sum(w.calls for w in ws)
    print("%d host thread(s), one context each, %d proofs per call: %6.0f proofs/s  (%.3f ms per call per thread, %d calls)"
          % (nthreads, k, calls * k / dt, dt / max(1, calls / nthreads) * 1e3, calls), flush=True)
    for w in ws:
        w.vk.close()
        w.eng.close()
