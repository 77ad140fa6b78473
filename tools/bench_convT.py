"""ConvTranspose2d(k=2,s=2) as GEMM + pixel-shuffle epilogue: the four image-decoder shapes at C2, output-channel tile forced through the tuning hook."""
import os, sys
os.environ["MS_CONV_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maxstyle_amd import ops
dev = torch.device("cuda:0")
for (N, Cin, Cout, H) in ((16, 128, 64, 16), (16, 64, 32, 32), (16, 32, 16, 64), (16, 16, 16, 128)):
    x = torch.randn(N, Cin, H, H, device=dev); w = torch.randn(Cin, Cout, 2, 2, device=dev) * 0.1; b = torch.randn(Cout, device=dev)
    wp = ops.pack_convT_weight(w)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    line = f"{(N, Cin, Cout, H)}:"
    for nt in (0, 1, 2, 4):
        os.environ["MS_CONV_FORCE_NT"] = str(nt)
        out = ops.conv2d(x, wp, b, Cout, 1, 1, epi_mode=2)
        err = float((out.double() - ref).norm() / ref.norm())
        for _ in range(3):
            ops.conv2d(x, wp, b, Cout, 1, 1, epi_mode=2, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d(x, wp, b, Cout, 1, 1, epi_mode=2, out=out)
        e1.record(); e1.synchronize()
        line += f"  nt={nt}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us (err {err:.1e})"
    print(line)
