#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code object inside libh2agg.so (VGPRs, scratch bytes, LDS bytes, spills), read from the
code object's metadata notes — what the library that ships actually contains, no rebuild.

    python tools/kernel_resources.py [pattern ...]            table of the kernels whose demangled name contains a pattern
    python tools/kernel_resources.py --check                  exit 1 unless the budgets below hold (tests/test_capi_symbols.py)

Budgets: every k_msm_accumulate_lean instantiation <= 128 VGPRs and no scratch (four waves per SIMD is what the kernel is
built for: if register pressure ever exceeds it the compiler spills silently — ADVICE r4)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "halo2-snark-aggregator_amd", "libh2agg.so")
LLVM = "/opt/rocm/lib/llvm/bin"

BUDGETS = [  # (name pattern, max VGPRs, max scratch bytes)
    ("k_msm_accumulate_lean", 128, 0),
    ("k_msm_final_lp", 128, 0),
    ("k_fb_partition", 128, 0),
    ("k_fb_bucket_sort", 128, 128),   # (a 1024-thread workgroup caps it at 128; a few loop invariants are parked in scratch around the staged passes)
]

# EVERY kernel of the code object is checked: a kernel with a scratch segment or spilled VGPRs that is not
# listed here fails the CPU suite, so a new kernel cannot pick up spills silently.  name -> (max scratch bytes, max spilled
# VGPRs, why it has a scratch segment).  "stack": no spills — the segment holds a dynamically indexed local array (a point's
# 36 limb words, a digit table), which the compiler cannot keep in registers; these are latency-chain tail kernels or one-off
# setup kernels, a handful of waves each.
SCRATCH_ALLOWED = {
    "k_bases_generate": (296, 0, "stack; one-off workload generation"),
    "k_bases_generate_comb": (48, 0, "stack; one-off"),
    "k_bases_shift": (48, 0, "stack; one-off level construction"),
    "k_comb_msm": (312, 0, "stack"),
    "k_comb_table_build": (80, 0, "stack; one-off per table"),
    "k_eval_tail": (312, 0, "stack"),
    "k_eval_tail_affine2": (312, 0, "stack"),
    "k_fb_bucket_sort": (128, 16, "1024-thread workgroup: 128-VGPR cap; loop invariants parked around the staged passes"),
    "k_fb_wsum": (600, 0, "stack; three one-wave workgroups per MSM"),
    "k_fr_batch_op": (48, 0, "stack"),
    "k_g1_batch_scalar_mul": (264, 0, "stack"),
    "k_g1_batch_scalar_mul_w4": (40, 0, "stack"),
    "k_g1_batch_to_affine": (352, 0, "stack: the batch inversion's prefix products"),
    "k_g1_sum": (312, 0, "stack"),
    "k_g1_sum_strided_affine": (344, 0, "stack"),
    "k_jac_to_mont_affine": (352, 0, "stack: the batch inversion's prefix products"),
    "k_msm_accumulate_big": (312, 0, "stack; over-long buckets only (skewed scalars)"),
    "k_msm_big_combine": (312, 0, "stack; over-long buckets only"),
    "k_msm_reduce_segments": (312, 0, "stack"),
    "k_msm_reduce_segments_par4": (344, 11, "stack + 11 spills at the 256-VGPR cap of a four-lane segment sum"),
    "k_msm_window_sum": (312, 0, "stack"),
    "k_msm_window_sum_par4": (312, 0, "stack"),
    "k_part_scatter": (36, 0, "stack"),
    "k_part_scatter_packed": (36, 0, "stack"),
    "k_part_scatter_staged": (36, 0, "stack"),
    "k_poseidon_transcript": (316, 150, "the device sponge (thousands of proofs): a 9-word state x 9 limbs per lane at one wave per SIMD"),
    "k_small_sort": (36, 0, "stack"),
    "k_tape_level": (48, 0, "stack"),
    "k_tape_run": (48, 0, "stack"),
    "k_tape_run_lds": (48, 0, "stack"),
}


def base_name(demangled):
    return re.sub(r"<.*$", "", demangled).strip()


def kernels(lib=LIB):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    out, cur, pending = [], None, {}
    for line in notes.split("\n"):
        m = re.match(r"\s*(?:- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k in ("agpr_count", "group_segment_fixed_size"):   # (a kernel's keys are sorted: these two precede its .name)
            pending[k] = int(v)
        elif k == "name" and v.startswith("_Z"):
            cur = {"name": v, **pending}
            pending = {}
            out.append(cur)
        elif cur is not None and k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
            cur[k] = int(v)
    names = subprocess.check_output(["c++filt"], input="\n".join(k["name"] for k in out), text=True).split("\n")
    for k, n in zip(out, names):
        k["demangled"] = re.sub(r"\(.*$", "", n).replace("h2agg::", "").replace("void ", "")
    return [k for k in out if "vgpr_count" in k]


def main():
    ks = kernels()
    if "--check" in sys.argv:
        bad = []
        for pat, vmax, smax in BUDGETS:
            hit = [k for k in ks if pat in k["demangled"]]
            if not hit:
                bad.append("no kernel matches " + pat)
            for k in hit:
                if k["vgpr_count"] > vmax or k["private_segment_fixed_size"] > smax or (smax == 0 and k.get("vgpr_spill_count", 0)):
                    bad.append("%s: %d VGPRs, %d B scratch" % (k["demangled"], k["vgpr_count"], k["private_segment_fixed_size"]))
        for k in ks:
            scratch, spills = k["private_segment_fixed_size"], k.get("vgpr_spill_count", 0)
            if not scratch and not spills:
                continue
            allowed = SCRATCH_ALLOWED.get(base_name(k["demangled"]))
            if allowed is None:
                bad.append("%s: %d B scratch, %d spilled VGPRs — not on the allow-list" % (k["demangled"], scratch, spills))
            elif scratch > allowed[0] or spills > allowed[1]:
                bad.append("%s: %d B scratch, %d spilled VGPRs — allowed %d B, %d" % (k["demangled"], scratch, spills, allowed[0], allowed[1]))
        have = {base_name(k["demangled"]) for k in ks}
        for name in ("k_msm_accumulate",):   # instantiations the product must not carry (measure build only)
            if name in have:
                bad.append(name + " is in the product code object (measure build only)")
        print("\n".join(bad) if bad else "kernel budgets hold (%d kernels in the code object, %d with an allowed scratch segment)"
              % (len(ks), sum(1 for k in ks if k["private_segment_fixed_size"])))
        sys.exit(1 if bad else 0)
    pats = [a for a in sys.argv[1:] if not a.startswith("--")]
    print("%-64s %5s %5s %8s %8s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS"))
    for k in sorted(ks, key=lambda k: k["demangled"]):
        if pats and not any(p in k["demangled"] for p in pats):
            continue
        print("%-64s %5d %5d %8d %8d" % (k["demangled"][:64], k["vgpr_count"], k["sgpr_count"], k["private_segment_fixed_size"],
                                         k["group_segment_fixed_size"]))


if __name__ == "__main__":
    main()
