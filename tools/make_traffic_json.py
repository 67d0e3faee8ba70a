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
    """rows `<kernel name> <counter> <n> <avg_value> <avg_dur_us>` of tools/rocpd_summary.py's PMC section (the kernel name may
    contain spaces: template arguments, a leading `void`)"""
    out = {}
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 5:
                continue
            name = " ".join(p[:-4])   # (the summary keeps the LAST 40 characters of a name: a leading `void ` may be cut)
            if name.endswith(kernel) or kernel.endswith(name):
                out[p[-4]] = (float(p[-2]), float(p[-1]))
    return out


def main(prefix):
    import bench
    k = "h2agg::k_msm_accumulate_lean<0, true>"   # (the whole-MSM instantiation, two chains in lock step; <1> / <2>: chained slices)
    fetch = counters(prefix + "_pmc_fetch.txt", k)["FETCH_SIZE"]
    write = counters(prefix + "_pmc_write.txt", k)["WRITE_SIZE"]
    sq = counters(prefix + "_pmc_sq.txt", k)
    insts, dur_us = sq["SQ_INSTS_VALU"]
    clk_hz = sq["GRBM_GUI_ACTIVE"][0] / 8.0 / (dur_us * 1e-6)          # summed over the 8 XCDs
    n, windows = 1 << 20, 16
    FOUR_CYCLE = 1467 + 144 + 82 + 20       # mads of 8M+2S with one merged reduction, mul_lo, 64-bit shifts, 64-bit adds
    wave_adds = n * windows / 64.0
    print(json.dumps({
        "kernel": "k_msm_accumulate_lean", "log2n": 20,
        "workload": "2^20-point MSM, c=16, one launch (bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie-leg --agg-proofs 0)",
        "csrc_sha": bench.csrc_sha(),
        "fetch_kb": fetch[0], "write_kb": write[0],
        "bytes_per_launch": int((fetch[0] + write[0]) * 1024),
        "correction": "none applied, by calibration (tools/ubench_gather.hip under rocprofv3 --pmc FETCH_SIZE, profiles/r02_sweeps.txt): per-lane gathers of whole 64-byte records \u2014 this kernel's reads of its bases \u2014 are counted 1.00x (1 GiB gathered from a 1-GiB table: FETCH_SIZE 1 051 229 KB; from a 64-MiB table: 979 964 KB, the difference being L2 hits), while a coalesced 16-B/lane stream is counted at exactly 1/2 (524 290 KB for 1 GiB), the x2 case of MI355X_MICROARCH.md.  The 4-byte run walks (64 MiB of entries, 4 % of the reads) are not calibrated.  WRITE_SIZE = 524 288 buckets x 144 B = 75.5 MB + partial lines (the lean kernel has no scratch).",
        "valu_insts": insts, "duration_us": dur_us, "shader_clock_hz": clk_hz,
        "wave_instructions_per_mixed_add": insts / wave_adds,
        "ideal_cpi": (FOUR_CYCLE * 4 + (insts / wave_adds - FOUR_CYCLE) * 2) / (insts / wave_adds),
        "measured_rate_cpi": None,
        "ideal_cpi_note": "cycles per wave-instruction per SIMD if the VALU never stalled: the kernel's mix at the issue rates of "
                          "the part (v_mad_u64_u32 / v_mul_lo_u32 / 64-bit shifts 4 cycles per wave64, 32-bit add / and / "
                          "alignbit 2: profiles/r01_ubench_instruction_rates.txt, which quotes them at the 2.4 GHz nominal clock; "
                          "at the ~2.03 GHz the part sustains under this load — GRBM_GUI_ACTIVE of this very pass — a multiply-add "
                          "is 4.0 cycles): the 1 467 + 144 + 82 + 20 = 1 713 four-cycle instructions of one insertion (ISA of "
                          "k_msm_accumulate_lean<0, true>, tools/isa_pressure.py) x 4, the rest of the MEASURED instructions "
                          "per insertion (wave_instructions_per_mixed_add) x 2",
        "source": "%s_pmc_{fetch,write,sq}.txt (rocprofv3 --pmc, separate passes, per-dispatch average, summed over the 8 XCDs)"
                  % os.path.basename(prefix),
    }, indent=1))


def batch(prefix, log2n=22, nbatch=16):
    """counter evidence for BASELINE.json configs[4]'s per-GPU share: 16 instance-column MSMs of 2^22 - 6 points over one table
    with fixed-base levels (tools/fixed_base_big.py 22 --fixed-only under rocprofv3 --pmc: the path bench.py's
    aggregate.config4_share runs on): bytes of the accumulation kernel per launch and per AGGREGATION"""
    import bench
    k = "h2agg::k_msm_accumulate_lean<0, true>"
    fetch = counters(prefix + "_pmc_batch_fetch.txt", k)["FETCH_SIZE"]
    write = counters(prefix + "_pmc_batch_write.txt", k)["WRITE_SIZE"]
    per_launch = (fetch[0] + write[0]) * 1024
    algo = 96.0 * ((1 << log2n) - 6)
    print(json.dumps({
        "kernel": "k_msm_accumulate_lean", "log2n": log2n, "batch": nbatch,
        "workload": "16 MSMs of 2^22 - 6 points over one 2^22-point table with fixed-base levels, c = 20 (tools/fixed_base_big.py 22 "
                    "--fixed-only), one launch per MSM: 13 insertions per point, every (level, point) gathered once from 3.25 GiB of levels",
        "csrc_sha": bench.csrc_sha(),
        "fetch_kb_per_launch": fetch[0], "write_kb_per_launch": write[0], "avg_launch_us": fetch[1],
        "bytes_per_launch": int(per_launch),
        "bytes_per_aggregation": int(per_launch * nbatch),
        "traffic_over_algorithmic": per_launch / algo,
        "source": "%s_pmc_batch_{fetch,write}.txt (rocprofv3 --pmc, separate passes, per-dispatch average, summed over the 8 XCDs)"
                  % os.path.basename(prefix),
    }, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "batch":
        batch(sys.argv[1])
    else:
        main(sys.argv[1])
