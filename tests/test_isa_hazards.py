"""Build-time guard for the shortlist kernels' inline-asm loads (no GPU needed: hipcc cross-compiles).

The tile loop issues its LDS fragment reads and a scalar table load from inline asm and completes them in a LATER asm
`s_waitcnt lgkmcnt(0)` tied to the destination registers (the compiler would otherwise drain the
LDS-DMA queue in front of every LDS access).  Nothing tells the compiler that a destination is not ready in between, so a
copy, a spill or a reuse it schedules there reads or clobbers a register whose load is still in flight.  This test compiles
the product configuration of prefilter_kernels.hip to gfx950 assembly and fails on any such instruction."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_use_of_asm_load_destinations_before_their_wait(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    import isa_asm_hazards as H
    out = tmp_path / "prefilter.s"
    src = os.path.join(ROOT, "ch-bin_amd", "csrc", "prefilter_kernels.hip")
    # (the Makefile's COMMON flags)
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    lines = []
    kernels, loads, hazards = H.scan(str(out), out=lines.append)
    assert kernels >= 20 and loads >= 200, (kernels, loads)     # (the scan really saw the kernels and their asm loads)
    assert hazards == 0, "\n".join(l for l in lines if "<-" in l)
