import os, sys
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from maxstyle_amd import ops
dev = torch.device("cuda:0")
def rel(a, b): return float((a.cpu().double() - b).abs().max() / (b.abs().max() + 1e-30))
cases = [(2, 16, 16, 64, 64), (1, 16, 16, 9, 128), (2, 1, 16, 30, 72), (2, 20, 24, 13, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64), (1, 8, 33, 7, 100)]
only = os.environ.get("CASE")
for ci, (N, Cin, Cout, H, W) in enumerate(cases):
    if only is not None and int(only) != ci: continue
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1; b = torch.randn(Cout, generator=g)
    wp = ops.pack_conv_weight(w.to(dev))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    print("case", ci, (N, Cin, Cout, H, W), "plain", flush=True)
    out = ops.conv2d(x.to(dev), wp, b.to(dev), Cout, 3, 1); torch.cuda.synchronize()
    print("   rel", rel(out, ref), flush=True)
    print("   stats", flush=True)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(x.to(dev), wp, b.to(dev), Cout, 3, 1, stats=stats); torch.cuda.synchronize()
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)); torch.cuda.synchronize()
    print("   rel", rel(out, ref), rel(coef[:, 2], ref.mean((0, 2, 3))), flush=True)
    cf = torch.randn(Cin, 4, generator=g); cfd = cf.to(dev)
    print("   pro1", flush=True)
    xa = F.leaky_relu(cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1), 0.2)
    o1 = ops.conv2d(x.to(dev), wp, b.to(dev), Cout, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2); torch.cuda.synchronize()
    print("   rel", rel(o1, F.conv2d(xa, w.double(), b.double(), padding=1)), flush=True)
    print("   pro2", flush=True)
    x2 = torch.randn(N, Cin, H, W, generator=g)
    xb = cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1) * x2.double() + cf[:, 2].double().view(1, -1, 1, 1)
    base = torch.randn(N, Cout, H, W, generator=g)
    o2 = ops.conv2d(x.to(dev), wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                    pro_cstride=4, in2=x2.to(dev), epi_mode=1, out=base.to(dev).clone()); torch.cuda.synchronize()
    print("   rel", rel(o2, F.conv2d(xb, w.double(), None, padding=1) + base.double()), flush=True)
