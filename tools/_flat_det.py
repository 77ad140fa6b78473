"""First call vs later calls of one C4 inner step, flat form on / off (one setting per process): bitwise, and each against the direct form (scratch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import r5_cases as R5
from maxstyle_amd.options import set_library_option
dev = torch.device("cuda:0")
which, flat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
set_library_option("conv.wino_flat", flat)
direct = R5.one_step_buffers(dev, which, False)
runs = [R5.one_step_buffers(dev, which, True) for _ in range(n)]
for i, b in enumerate(runs):
    c = R5.kink_census(b, direct)
    nd = sum(1 for k in runs[0] if k in b and not torch.equal(b[k], runs[0][k]))
    nd2 = sum(1 for k in runs[-1] if k in b and not torch.equal(b[k], runs[-1][k]))
    print(f"flat={flat} call {i}: vs direct forward {c['worst_forward_rel']:.2e} flips {c['flips']}; buffers differing from call 0: {nd}, from the last call: {nd2}", flush=True)
direct2 = R5.one_step_buffers(dev, which, False)
print("direct form, call 0 vs call 1: buffers differing", sum(1 for k in direct if not torch.equal(direct[k], direct2[k])), flush=True)
