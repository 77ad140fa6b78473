import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
N, Cin, Cout, H, W = 16, 64, 64, 320, 320
x = torch.randn(N, Cin, H, W, generator=g).to(dev)
wp = ops.pack_conv_weight((torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).to(dev)); b = torch.randn(Cout, generator=g).to(dev)
u = torch.randn(N, Cout, H, W, generator=g).to(dev); coef = torch.randn(Cout, 4, generator=g).to(dev); out = torch.empty_like(u)
kind = sys.argv[1] if len(sys.argv) > 1 else "tail"
for _ in range(6):
    if kind == "tail":
        check(lib.ms_conv1x1_bnres(x.data_ptr(), out.data_ptr(), wp.data_ptr(), b.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), coef.data_ptr(), 0.2, 0, st), "bnres")
    else:
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), b.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "conv")
torch.cuda.synchronize()
