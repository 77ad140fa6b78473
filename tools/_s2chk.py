import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maxstyle_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for (N, Cin, Cout, H, W) in [(16, 128, 128, 32, 32), (16, 64, 128, 64, 64), (16, 16, 16, 256, 256), (4, 128, 128, 8, 8), (16, 32, 32, 128, 128)]:
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05; b = torch.randn(Cout, generator=g) * 0.1
    cf = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.zeros(Cin), torch.zeros(Cin)], 1).contiguous().to(dev)
    pa, pb, _ = ops.coef_ptrs(cf)
    xd = x.to(dev); wp = ops.pack_conv_weight(w).to(dev); bd = b.to(dev)
    for pm in (0, 1):
        kw = dict(pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2) if pm else {}
        outs = [ops.conv2d(xd, wp, bd, Cout, 3, 2, **kw).clone() for _ in range(3)]
        xin = x.double()
        if pm:
            xin = F.leaky_relu(cf[:, 0].cpu().double().view(1, -1, 1, 1) * xin + cf[:, 1].cpu().double().view(1, -1, 1, 1), 0.2)
        ref = F.conv2d(xin, w.double(), b.double(), stride=2, padding=1)
        err = float((outs[0].double().cpu() - ref).abs().max() / ref.abs().max())
        print((N, Cin, Cout, H, W), "pro", pm, "err vs fp64 %.2e" % err, "deterministic", bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])))
