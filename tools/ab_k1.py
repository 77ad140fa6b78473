"""Isolated timing of the 1x1 convolutions of one inner step (config 4 / config 2 shapes) through the C ABI on random tensors: the streaming kernel (csrc/ms_conv_k1s.h)
against the tiled first-generation kernel (ms_conv_k1s_enable).   python tools/ab_k1.py [c4|c2] [reps]
Per shape: us per launch of each (median), fraction of 8 TB/s on the algorithmic bytes, whether the outputs have the same bits."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check

# (kind, Cin, Cout, H, W): kind "tail" = ms_conv1x1_bnres, "up2" = ... with the input at half resolution (H, W = input size), "plain" = ms_conv2d
SHAPES = {
    "c4": [("tail", 64, 64, 320, 320), ("up2", 64, 64, 160, 160), ("tail", 64, 128, 160, 160), ("tail", 128, 256, 80, 80), ("up2", 128, 64, 80, 80),
           ("convT", 64, 64, 160, 160), ("plain", 64, 64, 320, 320), ("plain", 128, 64, 160, 160), ("plain", 64, 64, 160, 160), ("plain", 64, 128, 80, 80), ("plain", 128, 256, 40, 40),
           # the channel-heavy levels (LDS-tiled GEMM form, csrc/ms_conv_k1g.h)
           ("tail", 256, 512, 40, 40), ("tail", 512, 512, 20, 20), ("plain", 512, 512, 20, 20), ("plain", 512, 256, 40, 40), ("plain", 256, 128, 80, 80), ("plain", 256, 512, 20, 20)],
    # (round 6) the 1x1 convolutions of config 2 on images of <= 32 x 32 pixels: the first-generation kernel's launches of the step (profiles/r05_step_budget_c2.txt)
    "c2small": [("tail", 64, 128, 32, 32), ("tail", 128, 128, 16, 16), ("plain", 128, 128, 16, 16), ("up2", 128, 64, 8, 8), ("up2", 64, 32, 16, 16), ("up2", 32, 16, 32, 32),
                ("plain", 32, 64, 32, 32), ("plain", 64, 128, 16, 16), ("plain", 128, 64, 32, 32)],
    "c2": [("tail", 16, 16, 256, 256), ("tail", 16, 32, 128, 128), ("tail", 32, 64, 64, 64), ("tail", 64, 128, 32, 32), ("up2", 32, 16, 64, 64), ("up2", 64, 32, 32, 32),
           ("plain", 16, 16, 256, 256), ("plain", 16, 16, 128, 128), ("plain", 32, 16, 128, 128), ("plain", 64, 32, 64, 64), ("plain", 16, 32, 64, 64)],
}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    N = 16
    g = torch.Generator().manual_seed(3)
    st = torch.cuda.current_stream().cuda_stream
    print(f"# {cfg}: us per launch (median of {reps}) and fraction of 8 TB/s: tiled | streaming")
    for kind, Cin, Cout, H, W in SHAPES[cfg]:
        x = torch.randn(N, Cin, H, W, generator=g).to(dev)
        wp = ops.pack_conv_weight((torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).to(dev))
        b = torch.randn(Cout, generator=g).to(dev)
        Ho, Wo = (2 * H, 2 * W) if kind in ("up2", "convT") else (H, W)
        if kind == "convT":
            wp = ops.pack_convT_weight((torch.randn(Cin, Cout, 2, 2, generator=g) * 0.2).to(dev))
        out = torch.empty(N, Cout, Ho, Wo, device=dev)
        nbytes = x.numel() * 4 + out.numel() * 4
        if kind == "convT":
            def run():
                check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), b.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 2, 0, st), "ms_conv2d(epi 2)")
        elif kind == "plain":
            def run():
                check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), b.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d")
        else:
            u = torch.randn(N, Cout, Ho, Wo, generator=g).to(dev)
            coef = torch.randn(Cout, 4, generator=g).to(dev)
            nbytes += u.numel() * 4

            def run():
                check(lib.ms_conv1x1_bnres(x.data_ptr(), out.data_ptr(), wp.data_ptr(), b.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), coef.data_ptr(), 0.2, 1 if kind == "up2" else 0, st), "bnres")
        line, outs = f"{kind:5s} {Cin:4d}->{Cout:4d} @{H}x{W}  {nbytes / 1e6:8.1f} MB ", []
        for on in (0, 1, 0, 1):
            lib.ms_set_option(b"conv.k1s", int(on)); lib.ms_set_option(b"conv.k1g", int(on))
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            us = ts[len(ts) // 2]
            outs.append(out.clone())
            line += f" | {'stream' if on else 'tiled '} {us:7.1f} us {nbytes / 8e12 / (us * 1e-6):.2f}"
        lib.ms_set_option(b"conv.k1s", 1); lib.ms_set_option(b"conv.k1g", 1)
        print(line + ("   same bits" if torch.equal(outs[0], outs[1]) else "   DIFFERENT BITS"), flush=True)


if __name__ == "__main__":
    main()
