#!/usr/bin/env python3
"""Empirical window-width table: for each MSM size time every (GLV mode, window) pair in ONE process and print
the best, so choose_window()'s rule can be checked against the hardware.  Run on the GPU box:
    python tools/window_sweep.py [--lo 10] [--hi 20] [--steps 20] [--overlap 2]
Throughput mode (--overlap 2, back-to-back MSMs) and latency mode (--overlap 0, sync after each) differ."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=10)
    ap.add_argument("--hi", type=int, default=20)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--overlap", type=int, default=2)
    ap.add_argument("--latency", action="store_true", help="synchronise after every MSM")
    args = ap.parse_args()
    pkg = entry.load_package()
    eng = pkg.H2Agg(0)
    dev = torch.device("cuda:0")
    _st = torch.cuda.Stream(dev); torch.cuda.set_stream(_st)   # (the default stream's handle is 0 = "the context's own stream")
    eng.set_stream(_st.cuda_stream)
    eng.msm_set_tail_overlap(args.overlap)
    rng = np.random.Generator(np.random.PCG64(7))
    nmax = 1 << args.hi
    d_k = torch.from_numpy(rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8)).to(dev)
    d_k[:, 31] &= 0x1f
    d_s = torch.from_numpy(rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8)).to(dev)
    d_s[:, 31] &= 0x1f
    table = eng.bases_generate(d_k.data_ptr(), nmax)
    d_out = torch.zeros((args.steps, 96), dtype=torch.uint8, device=dev)
    rows = []
    for lg in range(args.lo, args.hi + 1):
        n = 1 << lg
        res = {}
        for glv in (1, -1):                     # 1 = on, -1 = off
            m = 2 * n if glv == 1 else n
            c0 = max(3, min(16, m.bit_length() - 1 - 4))
            for c in range(max(3, c0 - 3), 17):
                eng.msm_configure(window_bits=c)
                eng.msm_configure_glv(glv)
                for i in range(3):
                    eng.g1_msm_device_async(table, d_s.data_ptr(), n, d_out[i].data_ptr())
                eng.synchronize(); torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(args.steps):
                    eng.g1_msm_device_async(table, d_s.data_ptr(), n, d_out[i].data_ptr())
                    if args.latency:
                        eng.synchronize()
                eng.synchronize(); torch.cuda.synchronize(dev)
                res[(glv, c)] = (time.perf_counter() - t0) / args.steps * 1e3
        best = min(res, key=res.get)
        eng.msm_configure(window_bits=0); eng.msm_configure_glv(0)
        for i in range(3):
            eng.g1_msm_device_async(table, d_s.data_ptr(), n, d_out[i].data_ptr())
        eng.synchronize(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(args.steps):
            eng.g1_msm_device_async(table, d_s.data_ptr(), n, d_out[i].data_ptr())
            if args.latency:
                eng.synchronize()
        eng.synchronize(); torch.cuda.synchronize(dev)
        auto = (time.perf_counter() - t0) / args.steps * 1e3
        line = "2^%d auto %.3f ms | best %s c=%d %.3f ms | " % (lg, auto, "glv" if best[0] == 1 else "plain", best[1], res[best])
        line += " ".join("%s%d=%.3f" % ("g" if g == 1 else "p", c, v) for (g, c), v in sorted(res.items()))
        print(line, flush=True)
        rows.append({"log2n": lg, "auto_ms": auto, "best": {"glv": best[0] == 1, "c": best[1], "ms": res[best]},
                     "all": {("g" if g == 1 else "p") + str(c): v for (g, c), v in res.items()}})
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
