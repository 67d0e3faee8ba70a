"""CPU: the liveness allocator behind the LDS register file of the Fr tape (csrc/schema_api.inc tape_lds_assign, kernel
csrc/schema.hpp k_tape_run_lds).  The two host functions are cut out of the PRODUCT source as they stand and compiled into
tests/cpp/tape_lds_driver.cpp, which simulates the kernel's level / barrier semantics on random programs: every register must
come out right, no slot may be written in a level that reads it, programs with more live values than slots must be refused
untouched.  (The kernel itself: tests/test_gpu_parity.py::test_fr_tape_register_file_in_lds.)"""
import os
import re
import shutil
import subprocess

import __graft_entry__ as entry


def test_tape_lds_allocator(tmp_path):
    csrc = os.path.join(entry.PKG_DIR, "csrc")
    api = open(os.path.join(csrc, "schema_api.inc")).read()
    i, j = api.index("bool schedule_levels("), api.index("// host half of eval()")
    (tmp_path / "tape_lds_extract.inc").write_text(api[i:j])
    hpp = open(os.path.join(csrc, "schema.hpp")).read()
    slots = re.search(r"constexpr uint32_t TAPE_LDS_SLOTS = (\d+);", hpp).group(1)
    noslot = re.search(r"constexpr uint32_t TAPE_NOSLOT = (0x[0-9a-f]+)u;", hpp).group(1)
    exe = str(tmp_path / "tape_lds_driver")
    subprocess.run([shutil.which("g++") or "g++", "-std=c++17", "-O2", "-Wall", "-I", str(tmp_path), "-DH2AGG_TAPE_LDS_SLOTS=%su" % slots,
                    "-DH2AGG_TAPE_NOSLOT=%su" % noslot, os.path.join(entry.ROOT, "tests", "cpp", "tape_lds_driver.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TAPE-LDS-ALLOC-OK" in r.stdout, r.stdout + r.stderr
