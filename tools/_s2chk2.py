import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maxstyle_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
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
