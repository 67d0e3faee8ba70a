"""A dozen default h2agg_verify_aggregation calls (4 proofs, pairing included) for a kernel / copy timeline of the LAST one
(tools/pipeline_trace.sh runs this under rocprofv3 --kernel-trace --memory-copy-trace with H2AGG_TRACE_PHASES=1)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
from bench import gen_scalars
pkg = entry.load_package()
eng = pkg.H2Agg(0)
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
_, gk = gen_scalars(7, 1 << 17)
g_table = eng.bases_generate(torch.from_numpy(gk.copy()).to(dev).data_ptr(), 1 << 17)
eng.bases_precompute(g_table)
pool = syn.point_pool(eng, 0xA66)
comp = eng.g1_batch_compress(b"".join(pool))
pool_c = [comp[32 * i:32 * i + 32] for i in range(len(pool))]
shape = syn.CircuitShape(17, 300, pool)
vk = ver.VerifyingKey(eng, ver.encode_vk(shape, lambda p: p))
fr = syn.fr_stream(0xF00D)
proofs = [([b"".join(fr() for _ in range(64))], shape.random_transcript(pool_c, 100 + i)) for i in range(2 * n)]
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
sets = [[(vk, "syn", g_table, proofs[:n])], [(vk, "syn", g_table, proofs[n:])]]
for r in range(12):
    ver.verify_aggregation(eng, sets[r & 1], g2, g2)
vk.close()
