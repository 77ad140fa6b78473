"""Weight-gradient kernel timings at the C2 layer shapes (run on the GPU box): TFLOP/s vs the 157.3 TF fp32-MFMA peak and
algorithmic GB/s (P + Q read once) vs HBM."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench_kernels import timeit


def main():
    from maxstyle_amd import ops
    dev = torch.device("cuda:0")
    B = 16
    # (name, Cout, Cin, H(out), W, ks, stride, ups, prologues)
    layers = [("u4.c3 16->16@256 3x3 +pro", 16, 16, 256, 256, 3, 1, 0, True), ("u4.c0 16->16@256 3x3", 16, 16, 256, 256, 3, 1, 0, False),
              ("seg.u4.c0 ups 16->16@256", 16, 16, 256, 256, 3, 1, 1, False), ("inc0 1->16@256", 16, 1, 256, 256, 3, 1, 0, False),
              ("d1.c3 32->32@128 +pro", 32, 32, 128, 128, 3, 1, 0, True), ("d1.c0 16->32@128", 32, 16, 128, 128, 3, 1, 0, False),
              ("d2.c3 64->64@64 +pro", 64, 64, 64, 64, 3, 1, 0, True), ("d3.c3 128->128@32 +pro", 128, 128, 32, 32, 3, 1, 0, True),
              ("d4.c3 128->128@16 +pro", 128, 128, 16, 16, 3, 1, 0, True), ("u4.ci 16->16@256 1x1", 16, 16, 256, 256, 1, 1, 0, False),
              ("d1.ci 16->32@128 1x1", 32, 16, 128, 128, 1, 1, 0, False), ("d1.down 16->16 s2 @128", 16, 16, 128, 128, 3, 2, 0, False),
              ("d3.down 64->64 s2 @32", 64, 64, 32, 32, 3, 2, 0, False)]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    out = {}
    for name, co, ci, H, W, ks, s, ups, pro in layers:
        if only and only not in name:
            continue
        hq, wq = (H * s, W * s) if not ups else (H // 2, W // 2)
        dy = torch.randn(B, co, H, W, device=dev)
        x = torch.randn(B, ci, hq, wq, device=dev)
        kw = {}
        if pro:
            kw = dict(p_bnbwd=(torch.randn(co, 4, device=dev), torch.randn(B, co, H, W, device=dev)), q_act=(torch.randn(ci, 4, device=dev), 0.2))
        t = timeit(lambda: ops.conv_wgrad(dy, x, ks, s, q_fetch=ups, **kw), 30)
        flops = 2.0 * B * H * W * co * ci * ks * ks
        byts = 4.0 * (dy.numel() * (2 if pro else 1) + x.numel())
        out[name] = {"us": round(t * 1e6, 1), "TFLOPs": round(flops / t / 1e12, 1), "GBps": round(byts / t / 1e9)}
    if only:
        for k, v in out.items():
            print(k.ljust(34), json.dumps(v))
        return
    # ConvTranspose 2x2 s2 (image decoder up4: 16->16, 128 -> 256)
    x = torch.randn(B, 16, 128, 128, device=dev); g = torch.randn(B, 16, 256, 256, device=dev)
    t = timeit(lambda: ops.conv_wgrad(x, g, 2, 2), 30)
    out["u4.up convT 16->16 128->256"] = {"us": round(t * 1e6, 1), "TFLOPs": round(2.0 * B * 128 * 128 * 16 * 16 * 4 / t / 1e12, 1), "GBps": round(4.0 * (x.numel() + g.numel()) / t / 1e9)}
    for k, v in out.items():
        print(k.ljust(34), json.dumps(v))


if __name__ == "__main__":
    main()
