"""Where should page-locked host buffers live?  h2agg_g1_msm from page-locked buffers first-touched on each NUMA node in turn.
    python tools/numa_pcie.py [log2n]"""
import ctypes, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
from bench import gen_scalars


def cpulist(s):
    out = set()
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
print("NUMA nodes:", {k: len(v) for k, v in nodes.items()})
for dev in glob.glob("/sys/bus/pci/devices/*"):
    try:
        if open(dev + "/vendor").read().strip() == "0x1002" and open(dev + "/class").read().strip().startswith(("0x0302", "0x1200", "0x0380")):
            print("GPU", os.path.basename(dev), "numa_node", open(dev + "/numa_node").read().strip())
    except OSError:
        pass
allowed = os.sched_getaffinity(0)
print("allowed cpus:", len(allowed), "running on cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
pkg = entry.load_package()
eng = pkg.H2Agg(0)
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
_, k_np = gen_scalars(1, n)
_, s_np = gen_scalars(2, n)
d_k = torch.from_numpy(k_np.copy()).cuda()
table = eng.bases_generate(d_k.data_ptr(), n)
bases = eng.bases_download(table, 0, n)
scal = bytes(s_np.tobytes())
for node, cpus in nodes.items():
    use = cpus & allowed
    if not use:
        continue
    os.sched_setaffinity(0, use)
    time.sleep(0.01)
    pb, ps = eng.host_alloc(64 * n), eng.host_alloc(32 * n)
    ctypes.memmove(pb, bases, 64 * n)
    ctypes.memmove(ps, scal, 32 * n)
    os.sched_setaffinity(0, allowed)
    eng.g1_msm(pb, ps, n)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        eng.g1_msm(pb, ps, n)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("buffers first-touched on node %d: %.2f ms per call (min %.2f)" % (node, ts[3] * 1e3, ts[0] * 1e3), flush=True)
    eng.host_free(pb)
    eng.host_free(ps)
