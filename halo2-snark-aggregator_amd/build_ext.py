"""Build libh2agg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python halo2-snark-aggregator_amd/build_ext.py [--force]
    python halo2-snark-aggregator_amd/build_ext.py --measure    -> tools/libh2agg_measure.so, built with -DH2AGG_MEASURE_KNOBS
        (the experiment switches of csrc/h2agg.hip `knob()`; for A/B runs: copy it over libh2agg.so on the GPU box.  The product
        build has none of them.)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libh2agg.so")
SOURCES = ["h2agg.hip"]
DEPS = ["h2agg.hip", "pairing.hpp", "fp.hpp", "fp_asm.inc", "g1.hpp", "batch_kernels.hpp", "sort_kernels.hpp", "fb_sort_kernels.hpp", "msm_kernels.hpp", "scalar_mul_kernels.hpp", "lp_kernels.hpp", "schema.hpp", "schema_api.inc", "comm.inc", "transcript.inc", "verifier.inc", "poseidon_kernels.hpp", "poseidon_host.hpp", "poseidon_sponge_host.hpp", "poseidon_ifma_host.hpp", "../../include/h2agg.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + FLAGS + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[h2agg] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def build_measure() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    out = os.path.join(os.path.dirname(HERE), "tools", "libh2agg_measure.so")
    cmd = [hipcc] + FLAGS + ["-DH2AGG_MEASURE_KNOBS", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    print("[h2agg] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--measure" in sys.argv:
        print(build_measure())
    else:
        build(force="--force" in sys.argv)
        print(OUT)
