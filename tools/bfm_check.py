"""bf16-MFMA mode of the wide conv kernel (ms_conv2d_bf16m) against fp64 on bf16-rounded operands.  python tools/bfm_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maxstyle_amd import ops
BF = torch.bfloat16
dev = torch.device("cuda:0")
def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
rb = lambda t: t.to(BF).to(torch.float32)
for (N, Cin, Cout, H, W) in [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64), (1, 8, 33, 8, 100), (2, 1, 16, 32, 64), (2, 128, 128, 16, 16), (2, 64, 64, 40, 40), (1, 20, 24, 9, 36), (2, 256, 64, 20, 20)]:
    x = rb(_rand((N, Cin, H, W), 1)); x2 = rb(_rand((N, Cin, H, W), 2)); w = _rand((Cout, Cin, 3, 3), 3, 0.1)
    cf = _rand((Cin, 4), 5); cfd = cf.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    wb = rb(w).double()
    a, bb, cc = cf[:, 0].double().view(1, -1, 1, 1), cf[:, 1].double().view(1, -1, 1, 1), cf[:, 2].double().view(1, -1, 1, 1)
    out = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, 3, 1, mfma_bf16=True).float().cpu().double()
    ref = F.conv2d(x.double(), wb, None, padding=1)
    print((N, Cin, Cout, H, W), "plain   max rel-to-max err", float((out - ref).abs().max() / ref.abs().max()))
    o1 = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2, mfma_bf16=True).float().cpu().double()
    r1 = F.conv2d(rb(F.leaky_relu(a * x.double() + bb, 0.2).float()).double(), wb, None, padding=1)
    print("          pro1    max rel-to-max err", float((o1 - r1).abs().max() / r1.abs().max()))
    o2 = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                    pro_cstride=4, in2=x2.to(dev).to(BF), mfma_bf16=True).float().cpu().double()
    r2 = F.conv2d(rb((a * x.double() + bb * x2.double() + cc).float()).double(), wb, None, padding=1)
    print("          pro2    max rel-to-max err", float((o2 - r2).abs().max() / r2.abs().max()))
