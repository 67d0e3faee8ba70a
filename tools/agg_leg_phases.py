"""Host-side phase times of bench.py's aggregate leg (instance-column MSMs + schema build + fold + evaluation), one GPU:
    python tools/agg_leg_phases.py [--proofs 4] [--instance-log2 17] [--reps 20]
Under rocprofv3 --kernel-trace --output-format csv, tools/eval_timeline.py DIR --last N shows the kernels of the last repetition."""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
from bench import gen_scalars
ap = argparse.ArgumentParser()
ap.add_argument("--proofs", type=int, default=4); ap.add_argument("--instance-log2", type=int, default=17)
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--lpb", type=int, default=0); ap.add_argument("--seg", type=int, default=0); ap.add_argument("--commitments", type=int, default=300)
args = ap.parse_args()
pkg = entry.load_package(); eng = pkg.H2Agg(0); eng.msm_set_tail_overlap(2)
agg = importlib.import_module(entry.PKG_NAME + ".aggregate"); mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
syn = importlib.import_module(entry.PKG_NAME + ".synthetic")
dev = torch.device("cuda", 0)
_, gk = gen_scalars(7, 1 << args.instance_log2)
g_table = eng.bases_generate(torch.from_numpy(gk.copy()).to(dev).data_ptr(), 1 << args.instance_log2)
try:
    eng.bases_precompute(g_table)
except Exception as ex:
    print("no fixed-base levels:", ex)
backend = agg.GpuBackend(pkg, eng)
pool = syn.point_pool(eng, 0xA66)
specs, lam = syn.make_proofs(pool, args.proofs, args.commitments)
n_inst = (1 << args.instance_log2) - 6
idx = list(range(args.proofs))
d_inst = torch.randint(0, 256, (args.proofs, n_inst, 32), dtype=torch.uint8, device=dev); d_inst[:, :, 31] &= 0x1F
d_out = torch.zeros((args.proofs, 96), dtype=torch.uint8, device=dev)
torch.cuda.synchronize(dev)
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
for rep in range(args.reps + 2):
    if rep == 2: T.clear()
    t00 = time.perf_counter()
    t0 = time.perf_counter(); b = backend.new_builder(); tick("new_builder", t0)
    if args.lpb: eng.msm_configure_lanes_per_bucket(args.lpb)
    if args.seg: eng.msm_configure(reduce_segment=args.seg)
    t0 = time.perf_counter(); eng.g1_msm_device_batch_async(g_table, d_inst.data_ptr(), n_inst, args.proofs, d_out.data_ptr()); tick("instance_msm_launch", t0)
    t0 = time.perf_counter()
    proofs, first = [], []
    for i in idx:
        proof, q0 = syn.build_proof(b, mo.MultiOpenProof, specs[i]); first.append(q0); proofs.append(proof)
    tick("build_proofs", t0)
    t0 = time.perf_counter(); local = agg.local_weighted_proof(b, proofs, idx, args.proofs, lam); tick("lambda_fold", t0)
    t0 = time.perf_counter(); backend.prepare(b, local); tick("prepare", t0)
    t0 = time.perf_counter(); aff = eng.g1_batch_to_affine_device(d_out.data_ptr(), args.proofs); tick("instance_wait+to_affine", t0)
    t0 = time.perf_counter()
    for j, q in enumerate(first): b.query_set_commitment(q, aff[64 * j:64 * j + 64])
    tick("patch", t0)
    t0 = time.perf_counter(); left, right = backend.evaluate(b, local); tick("evaluate", t0)
    tick("total", t00)
for k, v in T.items(): print("%-26s %8.3f ms" % (k, v / args.reps * 1e3))
