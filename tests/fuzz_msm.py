#!/usr/bin/env python3
"""Randomised differential run of the MSM against the C oracle (not part of the pytest suites: minutes of GPU time).
Random sizes, window widths, GLV on / off, lanes per bucket, sort knobs, reduce segment, overlap level and scalar patterns
(zeros, ones, r - 1, short scalars, repeated scalars, identity bases).  Usage on the GPU box:
    python tests/fuzz_msm.py [--seconds 120] [--seed 1] [--max-log2n 13]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from oracle import bn254 as O, cref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-log2n", type=int, default=13)
    args = ap.parse_args()
    pkg = entry.load_package()
    eng = pkg.H2Agg(0)
    rng = O.SplitMix64(args.seed)
    pool_n = 4096
    ks = [rng.fr() for _ in range(pool_n)]
    pool = eng.g1_batch_to_affine(eng.g1_batch_scalar_mul(O.aff_to_bytes(O.G1) * pool_n, b"".join(O.fe_to_bytes(k) for k in ks)))
    t_end = time.time() + args.seconds
    it = 0
    while time.time() < t_end:
        it += 1
        n = 1 + rng.next() % (1 << (1 + rng.next() % args.max_log2n))
        c = 0 if rng.next() % 4 == 0 else (16 if rng.next() % 3 == 0 else 2 + rng.next() % 15)   # 16: digit-major sort, 2-D reduction
        glv = (0, 1, -1)[rng.next() % 3]
        lpb = (0, 1, 2, 4, 8, 16)[rng.next() % 6]
        seg = (0, 1, 2, 8, 32)[rng.next() % 5]
        sub = (0, 4, 6, 9, 11)[rng.next() % 5]
        tile = (0, 0, 0, 256, 1024, -1, -2, -3)[rng.next() % 8]
        ovl = rng.next() % 4
        pat = rng.next() % 8
        idx = [rng.next() % pool_n for _ in range(n)]
        bases = bytearray(b"".join(pool[64 * i:64 * i + 64] for i in idx))
        if rng.next() % 5 == 0:
            for _ in range(1 + n // 7):
                j = rng.next() % n
                bases[64 * j:64 * j + 64] = bytes(64)                  # identity bases
        bases = bytes(bases)
        if pat == 0:
            sc = [0] * n
        elif pat == 1:
            sc = [1] * n
        elif pat == 2:
            sc = [O.R - 1] * n
        elif pat == 3:
            sc = [rng.next() % (1 << 16) for _ in range(n)]
        elif pat == 4:
            v = rng.fr()
            sc = [v] * n
        elif pat == 5:
            sc = [(O.R - 1 - rng.next() % 3) if rng.next() % 2 else rng.next() % 3 for _ in range(n)]
        else:
            sc = [rng.fr() for _ in range(n)]
        scal = b"".join(O.fe_to_bytes(s) for s in sc)
        cfg = dict(n=n, c=c, glv=glv, lpb=lpb, seg=seg, sub=sub, tile=tile, ovl=ovl, pat=pat, it=it)
        try:
            eng.msm_configure(window_bits=c, reduce_segment=seg)
            eng.msm_configure_glv(glv)
            eng.msm_configure_lanes_per_bucket(lpb)
            eng.msm_configure_sort(sub, tile)
            eng.msm_set_tail_overlap(ovl)
            got = eng.g1_batch_to_affine(eng.g1_msm(bases, scal))
        except pkg.H2AggError as e:
            if "sort" in str(e) or "window" in str(e) or "segment" in str(e) or "partition" in str(e) or "lanes" in str(e):
                continue                                               # knob combination rejected up front: fine
            print("ERROR", cfg, e)
            raise
        want = cref.msm_pippenger(bases, scal, n)
        if got != want:
            print("MISMATCH", cfg)
            sys.exit(1)
    print("fuzz ok: %d cases" % it)


if __name__ == "__main__":
    main()
