#!/usr/bin/env python3
"""HBM roofline rows for the batch kernels ("for the Fr / point batch kernels the HBM roofline is meaningful").

    rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python tools/batch_roofline.py run      (drives the kernels, n = 2^22)
    python tools/rocpd_summary.py /tmp/prof_b/b_results.db > gpurun_out/<tag>_batch_kernel_stats.txt
    python tools/batch_roofline.py report gpurun_out/<tag>_batch_kernel_stats.txt                    (GB/s against 8 TB/s)

Algorithmic bytes per element = the C-ABI operands in + results out (the kernels read / write exactly those buffers)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 1 << 22
HBM_PEAK_GBS = 8000.0
# kernel -> (bytes per element, what)
ROWS = {
    "h2agg::k_fr_batch_op": (96, "Fr a (32) + b (32) in, 32 out — averaged over ADD and MUL launches"),
    "h2agg::k_g1_batch_add": (288, "two Jacobian points in (2 x 96), one out (96)"),
    "h2agg::k_g1_batch_to_affine": (160, "Jacobian in (96), affine out (64); one safegcd inversion per point"),
    "h2agg::k_g1_batch_decompress": (96, "32-byte encoding in, affine out (64); one square root (252 squarings) per point"),
    "h2agg::k_bases_to_mont": (128, "canonical affine in (64), Montgomery affine out (64)"),
}


def run():
    import numpy as np
    import __graft_entry__ as entry
    pkg = entry.load_package()
    eng = pkg.H2Agg(0)
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = np.random.Generator(np.random.PCG64(5))
    a = rng.integers(0, 256, size=(N, 32), dtype=np.uint8)
    a[:, 31] &= 0x1F                                        # < 2^253 < r
    b = rng.integers(0, 256, size=(N, 32), dtype=np.uint8)
    b[:, 31] &= 0x1F
    ab, bb = a.tobytes(), b.tobytes()
    for _ in range(3):
        eng.fr_batch_op(pkg.OP_ADD, ab, bb)
        eng.fr_batch_op(pkg.OP_MUL, ab, bb)
    g = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    m = 4096
    pool = eng.g1_batch_scalar_mul(g * m, ab[:32 * m])       # 4096 Jacobian points, tiled to N
    jac = pool * (N // m)
    jac2 = (pool[96 * 17:] + pool[:96 * 17]) * (N // m)
    for _ in range(2):
        eng.g1_batch_add(jac, jac2)
        aff = eng.g1_batch_to_affine(jac)
    comp = eng.g1_batch_compress(aff)
    for _ in range(2):
        eng.g1_batch_decompress(comp)
    h = eng.bases_upload(aff)                                # k_bases_to_mont at N points
    eng.bases_free(h)
    print("ran batch kernels at n = 2^22")


def report(path):
    rows = {}
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) >= 7 and p[0] in ROWS:
                rows[p[0]] = (int(p[1]), float(p[3]))           # calls, avg_us
    print("# batch kernels at n = 2^22 elements: algorithmic bytes / rocprofv3 average kernel duration vs HBM peak (8 TB/s)")
    print("%-32s %6s %10s %10s %8s  %s" % ("kernel", "calls", "avg_us", "GB/s", "frac", "bytes per element"))
    for k, (bpe, what) in ROWS.items():
        if k not in rows:
            continue
        calls, avg_us = rows[k]
        gbs = bpe * N / (avg_us * 1e-6) / 1e9
        print("%-32s %6d %10.1f %10.1f %8.4f  %d: %s" % (k, calls, avg_us, gbs, gbs / HBM_PEAK_GBS, bpe, what))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
