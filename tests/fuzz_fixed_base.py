#!/usr/bin/env python3
"""Offline randomised differential run of the fixed-base path (h2agg_bases_precompute) against the C oracle: random table
sizes, level widths, prefix lengths, batch sizes and scalar patterns.  Not collected by pytest."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
from oracle import bn254 as O, cref

pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
rng = O.SplitMix64(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
cases = 0
while time.time() < t_end:
    big = rng.next() % 6 == 0                                  # now and then a table whose levels take the (level, point) sort (c = 20)
    n = (1 << 18) + 1 + rng.next() % (1 << 17) if big else 1 + rng.next() % (1 << (1 + rng.next() % 12))
    if big and rng.next() % 4 == 0:
        n = (1 << 20) + (1 << 18) + rng.next() % (1 << 20)     # enough level-1 tiles for the key-major shape of the over-long path
    cw = (0 if rng.next() % 2 else 20) if big else (0 if rng.next() % 3 == 0 else 4 + rng.next() % 17)
    ks = b"".join(O.fe_to_bytes(rng.fr()) for _ in range(n))
    d_k = torch.frombuffer(bytearray(ks), dtype=torch.uint8).to(dev)
    table = eng.bases_generate(d_k.data_ptr(), n)
    bases = eng.bases_download(table, 0, n)
    try:
        eng.bases_precompute(table, cw)
    except pkg.H2AggError:
        eng.bases_free(table); continue
    for _ in range(4):
        m = 1 + rng.next() % n
        B = 1 + rng.next() % 5
        pat = rng.next() % (10 if big else 5)
        rows = []
        for q in range(B):
            if pat == 0: sc = [rng.fr() for _ in range(m)]
            elif pat == 1: sc = [rng.next() % 3 for _ in range(m)]
            elif pat == 2: sc = [O.R - 1 - rng.next() % 2 for _ in range(m)]
            elif pat == 3: sc = [rng.next() % (1 << 20) for _ in range(m)]
            elif pat == 4: sc = [rng.fr() if rng.next() % 2 else 0 for _ in range(m)]
            # (big tables) what skews the (level, point) sort: small scalars, a carry into an empty top digit, digits from few partitions
            elif pat == 5: sc = [rng.next() for _ in range(m)]                                       # 64-bit
            elif pat == 6: sc = [rng.next() | (rng.next() << 64) for _ in range(m)]                  # 128-bit
            elif pat == 7: sc = [rng.fr() >> 14 for _ in range(m)]                                   # uniform below ~2^240
            elif pat == 8:
                nparts = 1 + rng.next() % 900
                sc = [sum(((v % nparts) * 256 + 1 + (v >> 12) % 256) << (20 * w) for w in range(12)) for v in (rng.next() for _ in range(m))]
            else: sc = [(rng.next() % (1 << 13) + 1) << 240 if rng.next() % 3 else rng.next() % 50 for _ in range(m)]   # top digit / tiny
            rows.append(b"".join(O.fe_to_bytes(s) for s in sc))
        d_s = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(dev)
        d_out = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
        eng.msm_set_tail_overlap(rng.next() % 3)
        eng.g1_msm_device_batch_async(table, d_s.data_ptr(), m, B, d_out.data_ptr())
        got = eng.g1_batch_to_affine_device(d_out.data_ptr(), B)
        for q in range(B):
            want = cref.msm_pippenger(bases[:64 * m], rows[q], m)
            if got[64 * q:64 * q + 64] != want:
                print("MISMATCH", dict(n=n, cw=cw, m=m, B=B, pat=pat, q=q)); sys.exit(1)
            cases += 1
    eng.msm_set_tail_overlap(0)
    eng.bases_free(table)
print("fixed-base fuzz ok:", cases, "MSMs")
