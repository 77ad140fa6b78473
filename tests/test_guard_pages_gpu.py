"""The op-level and whole-call GPU tests once more with GUARD PAGES behind (and in front of) every tensor the package allocates (tests/guard_pages.py): any kernel access
past the end of an operand is a GPU memory fault and the test process aborts.  Round 6 found one such read by accident - the Winograd appendix of a layer with an odd
number of 16-channel blocks, harmless until the weights ended their memory segment - this run looks for them on purpose."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONTROL = r'''
import sys
sys.path.insert(0, "tests")
import torch
import guard_pages
guard_pages.install()
from maxstyle_amd import ops
dev = torch.device("cuda:0")
C = 16 + 4096
st, parts = ops.conv_stats_buffer(1, C, 8, 8, dev); st.zero_()
gamma = ops.torch.ones(16, device=dev)                      # 64 bytes, the last 64 of a 2 MiB block - the kernel is told it holds C values
beta = ops.torch.zeros(C, device=dev)
coef = ops.torch.zeros(C, 4, device=dev)
ops.check(ops.lib.ms_bn_finalize(st.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, coef.data_ptr(), C, 0), "ms_bn_finalize")
torch.cuda.synchronize()
print("survived")
'''


def _env(mode):
    env = dict(os.environ)
    env["MS_GUARD_PAGES"] = mode
    env["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
    return env


def test_a_read_past_a_guarded_tensor_is_fatal():
    """Positive control: the guard works on this box.  ms_bn_finalize is handed a gamma vector of 16 values and told it holds 4112: the read of gamma[16] leaves the
    block, the process dies with a GPU memory fault (without the guard the same call reads whatever follows and returns)."""
    r = subprocess.run([sys.executable, "-c", CONTROL], cwd=ROOT, env=_env("end"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "survived" not in r.stdout, (r.returncode, r.stdout[-500:])
    assert "Memory access fault" in (r.stderr + r.stdout), r.stderr[-800:]


@pytest.mark.parametrize("mode", ["end", "start"])
def test_kernels_stay_inside_their_operands(mode):
    """Every convolution form (Winograd tiles / blocks / flat list with channel tails and odd block counts, second-generation narrow-row / streaming / stride-2 kernels, the
    random-shape sweep), the MaxStyle kernels, and one whole inner step at config 2 and config 4 (tests/test_round6_gpu.py::test_calls_are_bit_reproducible: every engine
    buffer, packed weight and table in its own guarded block) - one pytest process per file, each must finish."""
    files = ["tests/test_wino_gpu.py", "tests/test_conv_gpu.py", "tests/test_k3n_gpu.py", "tests/test_k1s_gpu.py", "tests/test_s2pro_gpu.py", "tests/test_gen2_random_gpu.py",
             "tests/test_maxstyle_gpu.py", "tests/test_round6_gpu.py"]
    if mode == "start":      # (the page in FRONT of a block: the Winograd family and the whole calls; `bash tools/guard_run.sh start` runs every file - clean at the end of round 6)
        files = ["tests/test_wino_gpu.py", "tests/test_k3n_gpu.py", "tests/test_round6_gpu.py"]
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "guard_run.sh"), mode] + files, cwd=ROOT, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT), capture_output=True, text=True,
                       timeout=3000)
    lines = [l for l in r.stdout.splitlines() if l.startswith("guard=")]
    assert len(lines) == len(files), r.stdout[-2000:] + r.stderr[-500:]
    bad = [l for l in r.stdout.splitlines() if "EXIT CODE" in l or "failed" in l or "Memory access fault" in l]
    assert not bad, r.stdout[-3000:]
