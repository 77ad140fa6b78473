import os, sys
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from maxstyle_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
N, Cin, Cout, H, W = 4, 16, 32, 64, 64
x = torch.randn(N, Cin, H, W, generator=g) * 2 + 0.5; w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1; b = torch.randn(Cout, generator=g)
ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
out = ops.conv2d(x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), Cout, 3, 1, stats=stats)
coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
mean = ref.mean((0, 2, 3)); var = ref.var((0, 2, 3), unbiased=False)
print("mean err", float((coef[:, 2] - mean).abs().max()), "invstd rel err", float(((coef[:, 3] - 1 / torch.sqrt(var + 1e-5)) * torch.sqrt(var + 1e-5)).abs().max()))
tab = stats.cpu().view(-1, 4)
print("slots", tab[0, 0].item(), "count sum ch0", float(tab[1:1 + int(tab[0, 0]), 0].sum()), "expected", N * H * W)
for (N, Cin, Cout, H, W) in [(4, 16, 16, 64, 64), (4, 32, 32, 64, 64), (4, 16, 32, 64, 64), (4, 64, 32, 64, 64), (2, 16, 16, 128, 128)]:
    x = torch.randn(N, Cin, H, W, generator=g) * 2 + 0.5; w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1; b = torch.randn(Cout, generator=g)
    cf = torch.randn(Cin, 4, generator=g); cfd = cf.to(dev)
    xa = F.leaky_relu(cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1), 0.2)
    ref = F.conv2d(xa, w.double(), b.double(), padding=1)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), Cout, 3, 1, stats=stats, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
    mean = ref.mean((0, 2, 3)); var = ref.var((0, 2, 3), unbiased=False)
    print((N, Cin, Cout, H, W), "out rel", float((out.cpu().double() - ref).abs().max() / ref.abs().max()), "mean err", float((coef[:, 2] - mean).abs().max()),
          "invstd rel err", float(((coef[:, 3] - 1 / torch.sqrt(var + 1e-5)) * torch.sqrt(var + 1e-5)).abs().max()))
