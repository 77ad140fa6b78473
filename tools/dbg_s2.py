import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
for (N, Cin, Cout, H, W) in [(16, 16, 16, 256, 256), (16, 32, 32, 128, 128), (16, 64, 64, 64, 64), (16, 128, 128, 32, 32), (16, 64, 64, 320, 320), (16, 512, 512, 40, 40)]:
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1; b = torch.randn(Cout, generator=g)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    ref = F.conv2d(x.double(), w.double().to(dev), b.double().to(dev), stride=2, padding=1)
    outs = []
    for on in (0, 1):
        lib.ms_conv_s2g2_enable(on)
        out = torch.full(ref.shape, float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 3, 2, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "conv")
        torch.cuda.synchronize(); outs.append(out)
    lib.ms_conv_s2g2_enable(1)
    for name, o in zip(("gen1", "gen2"), outs):
        e = (o.double() - ref).abs()
        bad = (e > 1e-4 * ref.abs().max()) | torch.isnan(o)
        print((N, Cin, Cout, H, W), name, "max err / max|ref| = %.2e" % float(e[~torch.isnan(e)].max() / ref.abs().max()), "bad elements:", int(bad.sum()), "nan:", int(torch.isnan(o).sum()),
              ("first bad: %s" % bad.nonzero()[:4].tolist()) if int(bad.sum()) else "")
