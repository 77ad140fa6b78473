"""Winograd mode of the wide conv kernel against fp64.  MS_CONV_WINO=2 python tools/wino_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maxstyle_amd import ops
dev = torch.device("cuda:0")
def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
def err(o, r):
    return float((o.cpu().double() - r).abs().max() / r.abs().max())
for (N, Cin, Cout, H, W) in [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64), (1, 8, 33, 10, 100), (2, 16, 16, 6, 72), (2, 64, 64, 32, 32), (2, 128, 64, 20, 40), (1, 16, 16, 9, 36), (16, 128, 128, 32, 32), (16, 16, 16, 256, 256)]:
    x = _rand((N, Cin, H, W), 1); x2 = _rand((N, Cin, H, W), 2); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4)
    cf = _rand((Cin, 4), 5); cfd = cf.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    a, bb, cc = (cf[:, i].double().view(1, -1, 1, 1) for i in range(3))
    xd, x2d = x.to(dev), x2.to(dev)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, stats=stats)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
    mean = ref.mean((0, 2, 3)); invstd = 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)
    print((N, Cin, Cout, H, W), "plain+stats", err(out, ref), "mean", float((coef[:, 2] - mean).abs().max()), "invstd", float((coef[:, 3] / invstd - 1).abs().max()))
    o1 = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
    print("      pro1", err(o1, F.conv2d(F.leaky_relu(a * x.double() + bb, 0.2), w.double(), None, padding=1)))
    base = _rand((N, Cout, H, W), 6)
    o2 = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                    pro_cstride=4, in2=x2d, epi_mode=1, out=base.to(dev).clone())
    print("      pro2+acc", err(o2, F.conv2d(a * x.double() + bb * x2.double() + cc, w.double(), None, padding=1) + base.double()))
    u = _rand((N, Cout, H, W), 24) + 0.3
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1)
    og, tab = ops.conv2d_actbwd(xd, wp, Cout, 3, u.to(dev), coef4.to(dev), 0.2)
    c4 = coef4.double()
    pre = c4[:, 0].view(1, -1, 1, 1) * u.double() + c4[:, 1].view(1, -1, 1, 1)
    refm = F.conv2d(x.double(), w.double(), None, padding=1) * torch.where(pre > 0, 1.0, 0.2)
    safe = (pre.abs() > 1e-4).double()
    bc = ops.bn_bwd_coefs(tab, 0, coef4.to(dev), N * H * W).cpu().double()
    s1 = refm.sum((0, 2, 3)); s2 = (refm * (u.double() - c4[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
    cnt = N * H * W
    be = -c4[:, 0] * (s2 * c4[:, 3] / cnt) * c4[:, 3]
    ref_bc = torch.stack([c4[:, 0], be, -c4[:, 0] * s1 / cnt - be * c4[:, 2]], 1)
    print("      actbwd", err(og.cpu().double() * safe, refm * safe), "table", float((bc[:, :3] - ref_bc).abs().max() / ref_bc.abs().max()))
torch.cuda.synchronize()
