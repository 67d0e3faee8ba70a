"""CPU tests: the C-ABI library loads, exports every symbol include/h2agg.h declares, and has no CPU mode."""
import ctypes
import os
import re

import pytest


def declared_symbols(header_path):
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(h2agg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(pkg):
    names = declared_symbols(pkg.HEADER_PATH)
    assert len(names) >= 25
    lib = ctypes.CDLL(pkg.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding declares a prototype for each of them
    assert sorted(pkg.exported_symbols()) == names


def test_no_cpu_mode(pkg):
    """Without a usable HIP device a context cannot be created: there is no CPU fallback to hide behind."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.H2AggError) as ei:
        pkg.H2Agg(0)
    assert ei.value.code == pkg.ERR_HIP
    lib = pkg.load_library()
    ctx = ctypes.c_void_p()
    assert lib.h2agg_create(-1, ctypes.byref(ctx)) == pkg.ERR_HIP and not ctx.value


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dirpath, _dirs, files in os.walk(os.path.join(root, "halo2-snark-aggregator_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<].*oracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_kernel_register_budgets_of_the_built_library():
    """The lean bucket accumulation is built for four waves per SIMD: 128 VGPRs, no scratch.  If register pressure ever
    exceeded that the compiler would spill silently, so the budgets are read off the code object that ships
    (tools/kernel_resources.py --check), here as well as on the GPU box.  The check walks EVERY kernel of the code object: one
    with a scratch segment or spilled VGPRs that is not on the tool's allow-list (name, bytes, spills, reason) fails, so a new
    kernel cannot pick up spills unseen.  The single-chain lean variants and the generic accumulation kernel are A/B builds
    only: they must not be in the product library."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "kernel_resources.py")
    r = subprocess.run([sys.executable, tool, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    table = subprocess.run([sys.executable, tool, "k_msm_accumulate_lean"], capture_output=True, text=True).stdout
    assert "k_msm_accumulate_lean<0, true>" in table and ", false>" not in table, table
    every = subprocess.run([sys.executable, tool, "k_msm_accumulate"], capture_output=True, text=True).stdout
    assert "k_msm_accumulate<" not in every, every
