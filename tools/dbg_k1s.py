import os, sys
sys.path.insert(0, "/root/repo")
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
for (N, Cin, Cout, H, W) in [(16, 64, 64, 320, 320), (16, 64, 64, 160, 160), (8, 64, 64, 64, 64), (16, 128, 64, 160, 160)]:
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    wp = ops.pack_conv_weight((torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).to(dev))
    outs = []
    for on in (0, 1):
        lib.ms_conv_k1s_enable(on)
        out = torch.full((N, Cout, H, W), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), 0, N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "conv")
        torch.cuda.synchronize()
        outs.append(out)
    lib.ms_conv_k1s_enable(1)
    d = (outs[0] != outs[1]) | torch.isnan(outs[1])
    print((N, Cin, Cout, H, W), "mismatching elements:", int(d.sum()), "of", d.numel(), "nan:", int(torch.isnan(outs[1]).sum()))
    if int(d.sum()):
        idx = d.nonzero()
        print(" first:", idx[:5].tolist(), " last:", idx[-3:].tolist())
        pn = d.flatten(2).any(1)          # [N, HW]
        units = pn.view(N, -1, 64).any(2) if (H * W) % 64 == 0 else None
        if units is not None:
            print(" bad units per image:", units.sum(1).tolist()[:8], " of", units.shape[1])
            bu = units[0].nonzero().flatten().tolist()
            print(" bad unit ids in image 0:", bu[:20])
        ch = d.flatten(2).any(2).any(0).nonzero().flatten().tolist()
        print(" bad channels:", ch[:20], len(ch))
        e = (outs[0] - outs[1]).abs()
        print(" max abs diff", float(e[~torch.isnan(e)].max()))
