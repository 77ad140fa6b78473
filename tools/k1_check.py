"""sha1 of the outputs of 1x1 convolutions on 16-pixel-wide tensors (4x16-pixel tiles) for an A/B of the chunk size (MS_CONV_K1N_CK) between two builds:
python tools/k1_check.py; MS_LIB=<alt build> python tools/k1_check.py - the lines must be identical."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
for (N, Cin, Cout, H, W) in [(16, 32, 64, 128, 128), (16, 64, 128, 64, 64), (16, 128, 256, 32, 32), (16, 128, 128, 16, 16), (16, 128, 64, 16, 16), (4, 512, 512, 4, 4), (16, 64, 128, 16, 16), (3, 100, 40, 8, 12), (16, 128, 128, 8, 8)]:
    x = torch.randn(N, Cin, H, W, generator=g).to(dev); w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1)
    b = torch.randn(Cout, generator=g).to(dev)
    cf = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.randn(Cin, generator=g) * 0.1, torch.zeros(Cin)], 1).contiguous().to(dev)
    pa, pb, pc = ops.coef_ptrs(cf)
    x2 = torch.randn(N, Cin, H, W, generator=g).to(dev)
    wp = ops.pack_conv_weight(w).to(dev)
    st, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    outs = [ops.conv2d(x, wp, b, Cout, 1, 1, stats=st),
            ops.conv2d(x, wp, b, Cout, 1, 1, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2),
            ops.conv2d(x, wp, None, Cout, 1, 1, pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2)]
    torch.cuda.synchronize()
    print((N, Cin, Cout, H, W), [hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:10] for o in outs], hashlib.sha1(st[1:].cpu().numpy().tobytes()).hexdigest()[:10])
