"""Micro-benchmark of single conv launches (for rocprofv3 --pmc / timing): python tools/bench_conv.py [case] [iters]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops

CASES = {
    # name: (N, Cin, Cout, H, W, ks, stride, in2)
    "c16_256": (16, 16, 16, 256, 256, 3, 1, False),
    "c16_256_bwd": (16, 16, 16, 256, 256, 3, 1, True),
    "c32_128": (16, 32, 32, 128, 128, 3, 1, False),
    "c64_64": (16, 64, 64, 64, 64, 3, 1, False),
    "c128_32": (16, 128, 128, 32, 32, 3, 1, False),
    "c128_16": (16, 128, 128, 16, 16, 3, 1, False),
    "c16_256_1x1": (16, 16, 16, 256, 256, 1, 1, False),
    "s2_16_256": (16, 16, 16, 256, 256, 3, 2, False),
    "s2_128_32": (16, 128, 128, 32, 32, 3, 2, False),
}


def run(name, iters):
    N, Cin, Cout, H, W, ks, stride, in2 = CASES[name]
    dev = torch.device("cuda:0")
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.1
    b = torch.randn(Cout, device=dev)
    wp = ops.pack_conv_weight(w)
    Ho, Wo = ops.conv_out_hw(H, W, ks, stride, 0)
    out = torch.empty(N, Cout, Ho, Wo, device=dev)
    stats, parts = ops.conv_stats_buffer(N, Cout, Ho, Wo, dev)
    kw = {}
    if in2:
        bc = torch.randn(Cin, 4, device=dev)
        pa, pb, pc = ops.coef_ptrs(bc)
        kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=torch.randn_like(x))
    fn = lambda: ops.conv2d(x, wp, b, Cout, ks, stride, out=out, stats=None if in2 else stats, **kw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / iters * 1e-3
    fl = 2.0 * N * Ho * Wo * Cout * Cin * ks * ks
    return {"case": name, "us": t * 1e6, "TFLOPs": fl / t / 1e12, "GBps_alg": (x.numel() + out.numel()) * 4 / t / 1e9}


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else list(CASES)
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for n in names:
        print(json.dumps(run(n, iters)))
