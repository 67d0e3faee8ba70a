"""Find the occasional 20-40 ms calls of h2agg_verify_aggregation: run many calls with H2AGG_TRACE_PHASES=1 and print the phase
lines of the slow ones.   python tools/stall_hunt.py [proofs] [calls]"""
import importlib, os, sys, time, subprocess
if os.environ.get("H2AGG_TRACE_PHASES") != "1":
    env = dict(os.environ, H2AGG_TRACE_PHASES="1")
    p = subprocess.run([sys.executable] + sys.argv, env=env, stderr=subprocess.STDOUT, stdout=subprocess.PIPE, text=True)
    lines = p.stdout.splitlines()
    slow = [i for i, l in enumerate(lines) if l.startswith("CALL") and float(l.split()[1]) > 8.0]
    print("calls:", sum(l.startswith("CALL") for l in lines), "slow:", len(slow))
    for i in slow[:12]:
        print(lines[i])
        j = i - 1
        while j >= 0 and not lines[j].startswith("CALL"):
            if "phases" in lines[j]:
                print("   ", lines[j])
            j -= 1
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
eng = pkg.H2Agg(0)
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
from bench import gen_scalars
nums = [a for a in sys.argv[1:] if a.isdigit()]
k = int(nums[0]) if nums else 4
calls = int(nums[1]) if len(nums) > 1 else 300
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
proofs = [([b"".join(fr() for _ in range(64))], shape.random_transcript(pool_c, 100 + i)) for i in range(k)]
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
arg = [(vk, "syn", g_table, proofs)]
ver.verify_aggregation(eng, arg, g2, g2)
if "--no-gc" in sys.argv:
    import gc
    gc.disable()
for i in range(calls):
    t0 = time.perf_counter()
    ver.verify_aggregation(eng, arg, g2, g2)
    sys.stderr.flush()
    print("CALL %.3f ms  #%d" % ((time.perf_counter() - t0) * 1e3, i), flush=True)
