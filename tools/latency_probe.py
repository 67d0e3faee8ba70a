"""Where the from-bytes call's time and its tail go: N calls of h2agg_verify_aggregation on 4 proofs with the
library's own phase split (h2agg_last_phases), per-phase p50 / p95 / max, the slowest calls' lines, how late the sponge
workers start after their chains are posted, and an A/B of debug keys.
    python tools/latency_probe.py [calls] [--ab key]        e.g. --ab tape_lds  (alternating blocks of 25 calls, key = 0 / 1)"""
import importlib, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
from bench import gen_scalars, host_noise
pkg = entry.load_package()
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
ver = importlib.import_module(entry.PKG_NAME + ".verifier")
nums = [a for a in sys.argv[1:] if a.isdigit()]
calls = int(nums[0]) if nums else 400
k = 4
ab = sys.argv[sys.argv.index("--ab") + 1] if "--ab" in sys.argv else None
dev = torch.device("cuda", 0)
_, gk = gen_scalars(7, 1 << 17)
d_gk = torch.from_numpy(gk.copy()).to(dev)
torch.cuda.synchronize()
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")


class Variant:
    """a context of its own per variant (A/B call by call: both see the same box at the same time)"""

    def __init__(self, value):
        self.eng = pkg.H2Agg(0)
        if ab:
            self.eng.debug_configure(ab, value)
        self.eng.debug_configure("phases", 1)
        self.g = self.eng.bases_generate(d_gk.data_ptr(), 1 << 17)
        self.eng.bases_precompute(self.g)
        pool = syn.point_pool(self.eng, 0xA66)
        comp = self.eng.g1_batch_compress(b"".join(pool))
        pool_c = [comp[32 * i:32 * i + 32] for i in range(len(pool))]
        shape = syn.CircuitShape(17, 300, pool)
        self.vk = ver.VerifyingKey(self.eng, ver.encode_vk(shape, lambda p: p))
        fr = syn.fr_stream(0xF00D)
        proofs = [([b"".join(fr() for _ in range(64))], shape.random_transcript(pool_c, 100 + i)) for i in range(2 * k)]
        self.sets = [[(self.vk, "syn", self.g, proofs[:k])], [(self.vk, "syn", self.g, proofs[k:])]]
        self.first = [ver.verify_aggregation(self.eng, a, g2, g2) for a in self.sets]
        for _ in range(4):
            for a in self.sets:
                ver.verify_aggregation(self.eng, a, g2, g2)


variants = [Variant(0), Variant(1)] if ab else [Variant(1)]
if ab:
    assert variants[0].first[0][:3] == variants[1].first[0][:3], "%s changes the result" % ab
import gc
gc.collect(); gc.freeze(); gc.disable()
rows = []
n0 = host_noise()
for i in range(calls):
    v = variants[i % len(variants)]
    b = (i // len(variants)) & 1
    t0 = time.perf_counter()
    got = ver.verify_aggregation(v.eng, v.sets[b], g2, g2)
    dt = time.perf_counter() - t0
    assert got[:3] == v.first[b][:3]
    rows.append((dt * 1e3, i % len(variants) if ab else 1, "#%d " % i + v.eng.last_phases()))
n1 = host_noise()


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def phases(line):
    d = {}
    for tok in line.split(" [")[0].split():
        if "=" in tok:
            a, b = tok.split("=")
            d[a] = d.get(a, 0.0) + float(b)
    return d


for variant in ((0, 1) if ab else (1,)):
    sel = [r for r in rows if r[1] == variant]
    ts = [r[0] for r in sel]
    tag = "%s=%d" % (ab, variant) if ab else "all"
    print("%-12s calls %4d  p50 %.3f  p95 %.3f  p99 %.3f  max %.3f ms   p95/p50 %.3f  max/p50 %.2f" % (
        tag, len(ts), pct(ts, .5), pct(ts, .95), pct(ts, .99), max(ts), pct(ts, .95) / pct(ts, .5), max(ts) / pct(ts, .5)))
    ph = [phases(r[2]) for r in sel]
    for name in ph[0]:
        v = [p.get(name, 0.0) for p in ph]
        print("    %-16s p50 %.3f  p95 %.3f  max %.3f" % (name, pct(v, .5), pct(v, .95), max(v)))
    late, runs = [], []
    for r in sel:
        m = re.search(r"posted at (\d+) us, start_us\+run_us@cpu:([^\]]*)\]", r[2])
        if m:
            posted = float(m.group(1))
            for tok in m.group(2).split():
                st, rest = tok.split("+")
                late.append(float(st) - posted)
                runs.append(float(rest.split("@")[0]))
    if late:
        print("    chains slower than 1.15 x the median chain: %d of %d" % (sum(1 for r in runs if r > 1.15 * pct(runs, .5)), len(runs)))
        print("    sponge chain start after posting: p50 %.0f  p95 %.0f  max %.0f us;  chain run time: p50 %.0f  p95 %.0f  max %.0f us" % (
            pct(late, .5), pct(late, .95), max(late), pct(runs, .5), pct(runs, .95), max(runs)))
print("host during the run:", {k_: n1[k_] - n0.get(k_, 0) for k_ in n1})
print("slowest calls:")
for dt, variant, line in sorted(rows, reverse=True)[:6]:
    print("  %.3f ms  (%s)  %s" % (dt, variant, line.strip()))
print("a median call:")
med = sorted(rows)[len(rows) // 2]
print("  %.3f ms  %s" % (med[0], med[2].strip()))
