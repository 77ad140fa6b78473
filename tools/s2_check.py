"""Op-level check of the stride-2 convolution kernels against fp64 (3x3 stride 2 with and without the BatchNorm-apply prologue; the 2x2 stride-2 data-gradient of
ConvTranspose2d) + run-to-run determinism: python tools/s2_check.py  (MS_LIB=<alternative build> for an A/B of compile-time choices, profiles/r03_experiments.txt 17)."""
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

# data-gradient of ConvTranspose2d(k2, s2): y = convT(x, w) -> dx = conv2d(dy, w, stride 2) with kernel 2
for (N, Cin, Cout, H, W) in [(16, 128, 128, 32, 32), (16, 64, 128, 64, 64), (16, 16, 32, 256, 256), (4, 128, 128, 16, 16), (16, 128, 64, 32, 32), (2, 20, 24, 12, 20)]:
    # convT weight [Cin_t, Cout_t, 2, 2] maps Cin_t (low-res) -> Cout_t (high-res); its data-gradient takes dy [N, Cout_t, H, W] -> dx [N, Cin_t, H/2, W/2]
    wt = torch.randn(Cout, Cin, 2, 2, generator=g) * 0.1          # Cin_t = Cout (of the dgrad), Cout_t = Cin (channels of dy)
    dy = torch.randn(N, Cin, H, W, generator=g)
    wp = ops.pack_convT_weight_dgrad(wt).to(dev)
    outs = [ops.conv2d(dy.to(dev), wp, None, Cout, 2, 2).clone() for _ in range(3)]
    ref = F.conv2d(dy.double(), wt.double(), stride=2)            # conv2d weight [out=Cin_t, in=Cout_t, 2, 2]
    err = float((outs[0].double().cpu() - ref).abs().max() / ref.abs().max())
    print((N, Cin, Cout, H, W), "k2s2 err vs fp64 %.2e" % err, "deterministic", bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])))
