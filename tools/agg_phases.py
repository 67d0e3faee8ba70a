#!/usr/bin/env python3
"""Phase timing of the aggregation path (schema build -> evaluate) for P proofs of C commitments on one GPU."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, importlib
import __graft_entry__ as entry

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ap = argparse.ArgumentParser(); ap.add_argument("--proofs", type=int, default=4); ap.add_argument("--commitments", type=int, default=280)
ap.add_argument("--reps", type=int, default=10); ap.add_argument("--overlap", type=int, default=2)
ap.add_argument("--window", type=int, default=0, help="force the Pippenger window of every MSM (0 = the measured table)")
args = ap.parse_args()
pkg = entry.load_package(); eng = pkg.H2Agg(0)
if args.overlap: eng.msm_set_tail_overlap(args.overlap)
if args.window: eng.msm_configure(args.window, 0, 0)
agg = importlib.import_module(entry.PKG_NAME + ".aggregate"); mo = importlib.import_module(entry.PKG_NAME + ".multiopen")
rng = np.random.Generator(np.random.PCG64(1))
fr = lambda: (int.from_bytes(rng.bytes(64), "little") % R_MOD).to_bytes(32, "little")
g_aff = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
pool = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(g_aff * 256, b"".join(fr() for _ in range(256))))
pts = [pool[64 * i:64 * i + 64] for i in range(256)]
lam = fr(); packed = []
for i in range(args.proofs):
    x, xw, xl = fr(), fr(), fr()
    qs = [(0, "p%d_instance_commitments0" % i, x)] + [(0, "p%d_advice_commitments%d" % (i, c), x) for c in range(args.commitments)]
    qs += [(1, "p%d_advice_commitments%d" % (i, c), xw) for c in range(0, args.commitments, 7)] + [(-6, "p%d_perm%d" % (i, c), xl) for c in range(3)]
    import ctypes as C
    # keys / rotations as C arrays built once per circuit (what synthetic.build_proof caches on the ProofSpec): the timed
    # path then only hands pointers over, as a Rust caller would
    packed.append((pkg.SchemaBuilder.keys_array([k for _r, k, _z in qs]) if not os.environ.get("FINE") else [k for _r, k, _z in qs],
                   b"".join(pts[(i * 131 + k) % 256] for k in range(len(qs))), b"".join(fr() for _ in qs),
                   (C.c_int32 * len(qs))(*[r for r, _k, _z in qs]), b"".join(z for _r, _k, z in qs), b"".join(pts[(i + j) % 256] for j in (1, 2, 3)), fr(), fr()))
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
for rep in range(args.reps + 1):
    if rep == 1: T.clear()
    t0 = time.perf_counter(); b = pkg.SchemaBuilder(eng); tick("builder_create", t0)
    proofs = []
    for i in range(args.proofs):
        keys, cm, ev, rots, zs, wb, v, u = packed[i]
        if os.environ.get("FINE"):
            import ctypes as C
            n = len(keys)
            t0 = time.perf_counter(); arr = (C.c_char_p * n)(*[k.encode() for k in keys]); out = (C.c_uint32 * n)(); tick("eq.encode", t0)
            t0 = time.perf_counter(); eng._check(b._lib.h2agg_schema_evaluation_queries(b._s, n, arr, cm, ev, out)); tick("eq.ccall", t0)
            t0 = time.perf_counter(); qn = [pkg.EvaluationQuerySchema(b, out[i]) for i in range(n)]; tick("eq.wrap", t0)
        else:
            t0 = time.perf_counter(); qn = b.evaluation_queries(keys, cm, ev, wrap=False); tick("evaluation_queries", t0)
        t0 = time.perf_counter(); wx, wg = b.batch_multi_open("p%d" % i, rots, zs, qn, wb, v, u); tick("batch_multi_open", t0)
        proofs.append(mo.MultiOpenProof(wx, wg))
    t0 = time.perf_counter(); local = agg.local_weighted_proof(b, proofs, list(range(args.proofs)), args.proofs, lam); tick("lambda_fold", t0)
    t0 = time.perf_counter(); l, r, _ = b.evaluate_multiopen_proof(local.w_x, local.w_g); tick("evaluate", t0)
    t0 = time.perf_counter(); b.close(); tick("close", t0)
tot = sum(T.values())
for k, v in T.items(): print("%-20s %8.3f ms" % (k, v / args.reps * 1e3))
print("%-20s %8.3f ms  (%d proofs x %d queries)" % ("total", tot / args.reps * 1e3, args.proofs, len(packed[0][0])))
