"""Small-image resampling convolutions of the reference's shipped workloads (20 x 1 x 192 x 192: levels of 12 / 24 / 48 pixels; 20 x 1 x 224 x 224: 14 / 28 / 56) and of
config 2 (16 / 32): the fused-fetch first-generation conv (ms_conv2d with fetch 1 = nearest up-sampling / 2 = zero insertion) against the sub-pixel kernel's second
generation in its two geometries (8 x 32-pixel tiles | sixteen 4 x 4-pixel blocks), isolated, event-timed (GPU box):  python tools/ab_subpix_small.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(5)
st = torch.cuda.current_stream().cuda_stream
TILES, BLOCKS = 2, 4
SHAPES = [("dg", 20, 128, 12, 128), ("dg", 20, 128, 14, 128), ("dg", 16, 128, 16, 128), ("dg", 20, 64, 24, 64), ("dg", 20, 64, 28, 64), ("dg", 16, 64, 32, 64), ("dg", 20, 32, 48, 32), ("dg", 20, 32, 56, 32),
          ("up", 20, 128, 12, 64), ("up", 20, 128, 14, 64), ("up", 16, 128, 16, 64), ("up", 20, 64, 24, 32), ("up", 20, 64, 28, 32), ("up", 16, 64, 32, 32), ("up", 20, 32, 48, 16), ("up", 20, 32, 56, 16)]


def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000.0)
    ts.sort()
    return ts[len(ts) // 2]


print("# kind N Cin Hs Cout : fused-fetch first generation | sub-pixel tiles | sub-pixel blocks | automatic   (us per launch, isolated; '-' = not legal)   max |blocks - fused| / range")
for kind, N, Cin, Hs, Cout in SHAPES:
    x = torch.randn(N, Cin, Hs, Hs, generator=g).to(dev)
    out = torch.empty(N, Cout, 2 * Hs, 2 * Hs, device=dev)
    if kind == "up":
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(dev)
        wp = ops.pack_conv_weight(w)
        sums = torch.empty(int(lib.ms_subpix_pack_floats(Cin, Cout)), device=dev)
        check(lib.ms_subpix_pack(wp.data_ptr(), sums.data_ptr(), Cin, Cout, st), "pack")
        stats = torch.zeros(int(lib.ms_conv_stats_bytes(N, Cout, 2 * Hs, 2 * Hs)) // 4, device=dev)
        fused = lambda: ops.conv2d(x, wp, None, Cout, 3, 1, fetch=ops.FETCH_UPS2, stats=stats, out=out)
        sub = lambda fl: check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), sums.data_ptr(), 0, N, Cin, Hs, Hs, Cout, 0, stats.data_ptr(), 0, 0, 0, 1.0, 0, fl, st), "subpix2")
    else:
        w = (torch.randn(Cin, Cout, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(dev)
        wp = ops.pack_conv_weight_dgrad(w)
        fused = lambda: ops.conv2d(x, wp, None, Cout, 3, 1, fetch=ops.FETCH_ZINS2, out=out)
        sub = lambda fl: check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), 0, 0, N, Cin, Hs, Hs, Cout, 1, 0, 0, 0, 0, 1.0, 0, fl, st), "subpix2")
    t_f = timeit(fused); ref = out.clone()
    legal_t = Hs % 4 == 0
    t_t = timeit(lambda: sub(TILES)) if legal_t else None
    t_b = timeit(lambda: sub(BLOCKS)); d = float((out - ref).abs().max() / ref.abs().max())
    t_a = timeit(lambda: sub(0))
    print(f"{kind} {N:3d} {Cin:4d} {Hs:3d} {Cout:4d} : {t_f:7.1f} | {('%7.1f' % t_t) if t_t else '      -'} | {t_b:7.1f} | {t_a:7.1f}    {d:.1e}")
