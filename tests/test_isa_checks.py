"""Static checks of the built ISA (CPU: hipcc cross-compiles gfx950 here).

The Winograd kernels' staging waves wait for the weights' LDS-DMA with a COUNTED s_waitcnt (vmcnt(kDataLoads): the next chunk's activation loads stay in flight).  The
count is a source-level constant; the loads are the compiler's.  Round 6: in the flat form the halo loads were provably dead, the compiler removed them, and the wait
stopped covering the last DMA pieces - a run-to-run difference of ~1e-5 in one work item, seen only on cold first calls.  tools/check_dma_wait.py counts the loads in the
ISA; this test keeps the two in step for the two-block instantiations (flat form, tiled form)."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maxstyle_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_counted_wait_behind_the_weights_dma_covers_it():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dma_wait
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in ("winof", "wino2"):
            out = os.path.join(tmp, f + ".s")
            cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-function", "-Wno-inline-asm",
                   "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only", os.path.join(CSRC, f"ms_conv_inst_{f}.hip"), "-o", out]
            procs.append((out, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        for out, p in procs:
            _, err = p.communicate(timeout=900)
            assert p.returncode == 0, err.decode()[-2000:]
            seen, bad = check_dma_wait.check(out)
            assert seen >= 20, (out, seen)          # the check found the waits it is about (the flags of the Makefile: the same code as the shipped library)
            assert not bad, bad
