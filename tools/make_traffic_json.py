#!/usr/bin/env python3
"""PMC evidence of the dominant kernel -> one JSON that bench.py reads (roofline.traffic, roofline.valu).

    python tools/make_traffic_json.py gpurun_out/<tag> > gpurun_out/<tag>_traffic.json

Inputs: <tag>_pmc_fetch.txt, <tag>_pmc_write.txt, <tag>_pmc_sq.txt as written by tools/profile_round.sh (rocprofv3 --pmc,
one counter group per pass, never combined with other traces).  The JSON carries the hash of csrc/ at collection time:
bench.py refuses to quote it for different kernel sources.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counters(path, kernel):
    out = {}
    with open(path) as f:
        for line in f:
            p = line.split()
            if p and p[0] == "void":        # (template instantiations are printed with their return type)
                p = p[1:]
            if len(p) == 5 and p[0] == kernel:
                out[p[1]] = (float(p[3]), float(p[4]))
    return out


def main(prefix):
    import bench
    k = "h2agg::k_msm_accumulate<0>"   # (the whole-MSM instantiation; <1> / <2> are the chained slices of the host-buffer path)
    fetch = counters(prefix + "_pmc_fetch.txt", k)["FETCH_SIZE"]
    write = counters(prefix + "_pmc_write.txt", k)["WRITE_SIZE"]
    sq = counters(prefix + "_pmc_sq.txt", k)
    insts, dur_us = sq["SQ_INSTS_VALU"]
    clk_hz = sq["GRBM_GUI_ACTIVE"][0] / 8.0 / (dur_us * 1e-6)          # summed over the 8 XCDs
    n, windows = 1 << 20, 16
    wave_adds = n * windows / 64.0
    print(json.dumps({
        "kernel": "k_msm_accumulate", "log2n": 20,
        "workload": "2^20-point MSM, c=16, one launch (bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie-leg --agg-proofs 0)",
        "csrc_sha": bench.csrc_sha(),
        "fetch_kb": fetch[0], "write_kb": write[0],
        "bytes_per_launch": int((fetch[0] + write[0]) * 1024),
        "correction": "none applied, by calibration (tools/ubench_gather.hip under rocprofv3 --pmc FETCH_SIZE, profiles/r02_sweeps.txt): per-lane gathers of whole 64-byte records \u2014 this kernel's reads of its bases \u2014 are counted 1.00x (1 GiB gathered from a 1-GiB table: FETCH_SIZE 1 051 229 KB; from a 64-MiB table: 979 964 KB, the difference being L2 hits), while a coalesced 16-B/lane stream is counted at exactly 1/2 (524 290 KB for 1 GiB), the x2 case of MI355X_MICROARCH.md.  The 4-byte run walks (64 MiB of entries, 4 % of the reads) are not calibrated.  WRITE_SIZE matches 524 288 buckets x 144 B = 75.5 MB + partial lines + 36 MB of scratch from the peeled second insertion.",
        "valu_insts": insts, "duration_us": dur_us, "shader_clock_hz": clk_hz,
        "wave_instructions_per_mixed_add": insts / wave_adds,
        "ideal_cpi": 3.57,
        "measured_rate_cpi": 4.25,
        "measured_rate_note": "the same mix with the multiply-add at the ~5 cycles per wave64 it takes on this part at even "
                              "wave counts (tools/ubench_chain.hip, profiles/r02_sweeps.txt): (1 467 x 5 + 226 x 4 + 467 x 2) / 2 160",
        "ideal_cpi_note": "cycles per wave-instruction per SIMD if the VALU never stalled: the kernel's mix at the measured "
                          "issue rates (v_mad_u64_u32 / v_mul_lo_u32 / v_lshl_add_u64 / 64-bit shifts 4 cycles per wave64, "
                          "32-bit add / and / cndmask 2; profiles/r01_ubench_instruction_rates.txt): 1 693 of 2 160 "
                          "VALU instructions per mixed addition are half rate (1 467 multiply-adds, 144 64-bit shifts, 82 "
                          "v_mul_lo) since the merges of partial sums went away (fp_mont_chain2)",
        "source": "%s_pmc_{fetch,write,sq}.txt (rocprofv3 --pmc, separate passes, per-dispatch average, summed over the 8 XCDs)"
                  % os.path.basename(prefix),
    }, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
