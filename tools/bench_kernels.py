"""Micro-benchmarks of individual HIP kernels (run on the GPU box): achieved algorithmic GB/s vs the 8 TB/s HBM peak."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default=None, help="comma-separated layer names (L3,L4,L5,C4_L4)")
    args = ap.parse_args()
    from maxstyle_amd import MaxStyle, ops
    dev = torch.device("cuda:0")
    out = {}
    # device-to-device copy as the practical HBM ceiling on this box
    a = torch.empty(64 * 1024 * 1024, device=dev); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a), args.iters)
    out["copy_256MB_GBps"] = 2 * a.numel() * 4 / t / 1e9
    shapes = {"L3": (16, 16, 128, 128), "L4": (16, 16, 256, 256), "L5": (16, 1, 256, 256), "C4_L3": (16, 64, 160, 160), "C4_L4": (16, 64, 320, 320),
              "C4_L5": (16, 3, 320, 320)}
    if args.only:
        shapes = {k: v for k, v in shapes.items() if k in args.only.split(",")}
    for name, shape in shapes.items():
        B, C, H, W = shape
        x = torch.randn(shape, device=dev)
        dy = torch.randn(shape, device=dev)
        layer = MaxStyle(B, C, p=1.5)
        y = layer(x)
        mu, sig = layer._last_stats
        perm = layer._perm_device(dev)
        gs, bs = layer.gamma_std, layer.beta_std
        lm, gn, bn = layer.lmda.detach(), layer.gamma_noise.detach(), layer.beta_noise.detach()
        yb = torch.empty_like(x)
        n = x.numel()
        t_f = timeit(lambda: ops.style_fwd(x, perm, lm, gn, bn, gs, bs, False, out=yb), args.iters)
        _, mu_, sig_, cA, cS = ops.style_fwd(x, perm, lm, gn, bn, gs, bs, False, out=yb)
        t_b = timeit(lambda: ops.style_bwd(dy, x, mu_, sig_, cA, gs, bs, lm, perm, True, True, True), args.iters)
        t_b0 = timeit(lambda: ops.style_bwd(dy, x, mu_, sig_, cA, gs, bs, lm, perm, False, True, True), args.iters)
        out[name] = {"shape": shape, "fwd_us": t_f * 1e6, "fwd_GBps": 8 * n / t_f / 1e9, "bwd_dx_us": t_b * 1e6, "bwd_dx_GBps": 12 * n / t_b / 1e9,
                     "bwd_nodx_us": t_b0 * 1e6, "bwd_nodx_GBps": 8 * n / t_b0 / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
