"""Isolated timing of the sub-pixel resampling convolutions (ms_conv_subpix) and the stride-2 3x3 forward (ms_conv2d) at the shapes of one inner step of
config 4 / config 2, straight through the C ABI on random tensors (GPU box):
    python tools/ab_subpix.py [c4|c2] [reps] [only-index]
Prints per shape: microseconds per launch (median of `reps` event-timed launches behind 3 warm-ups), the executed-MFMA fraction (executed flop / 157.3 TFLOP/s /
time) and a checksum of the output (sum of the fp32 bit patterns: equal checksums <=> equal bits, for A/B builds selected with MS_LIB / environment switches)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check

PEAK = 157.3e12

SHAPES = {
    "c4": [("up", 512, 20, 256), ("up", 256, 40, 128), ("up", 128, 80, 64), ("up", 64, 160, 64),
           ("dg", 512, 20, 512), ("dg", 256, 40, 256), ("dg", 128, 80, 128), ("dg", 64, 160, 64),
           ("dgm", 512, 20, 512), ("dgm", 256, 40, 256), ("dgm", 128, 80, 128), ("dgm", 64, 160, 64),
           ("s2", 512, 40, 512), ("s2", 256, 80, 256), ("s2", 128, 160, 128), ("s2", 64, 320, 64)],
    "c2": [("up", 128, 16, 64), ("up", 64, 32, 32), ("up", 32, 64, 16), ("up", 16, 128, 16),
           ("dg", 128, 16, 128), ("dg", 64, 32, 64), ("dg", 32, 64, 32), ("dg", 16, 128, 16),
           ("dgm", 128, 16, 128), ("dgm", 64, 32, 64), ("dgm", 32, 64, 32), ("dgm", 16, 128, 16),
           ("s2", 128, 32, 128), ("s2", 64, 64, 64), ("s2", 32, 128, 32), ("s2", 16, 256, 16)],
}


def bits_sum(t):
    return int(t.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    dev = torch.device("cuda:0")
    N = 16
    g = torch.Generator(device="cpu").manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    print(f"# {cfg}: us per launch (median of {reps}), executed fraction of the fp32 MFMA peak, output checksum")
    for idx, (kind, Cin, Hs, Cout) in enumerate(SHAPES[cfg]):
        if only >= 0 and idx != only:
            continue
        Ws = Hs
        x = torch.randn(N, Cin, Hs, Ws, generator=g).to(dev)
        if kind == "s2":
            w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (3.0 * Cin ** 0.5))).to(dev)
            wp = ops.pack_conv_weight(w)
            Ho = Hs // 2
            out = torch.empty(N, Cout, Ho, Ho, device=dev)
            stats = torch.zeros(int(lib.ms_conv_stats_bytes(N, Cout, Ho, Ho)) // 4, device=dev)
            flop = 2.0 * N * Ho * Ho * Cout * Cin * 9

            def run(flags=0):
                check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), 0, N, Cin, Hs, Ws, Cout, 3, 2, 0, 0, 0, 0, 0, 0, 4, 1.0, 0, 0, st), "ms_conv2d")      # (as the engine calls it: bias-free here, no statistics)
        else:
            mode = 0 if kind == "up" else 1
            Ho = 2 * Hs
            out = torch.empty(N, Cout, Ho, Ho, device=dev)
            if mode == 0:
                w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (3.0 * Cin ** 0.5))).to(dev)
                wp = ops.pack_conv_weight(w)
                stats = torch.zeros(int(lib.ms_conv_stats_bytes(N, Cout, Ho, Ho)) // 4, device=dev)
                flop = 2.0 * N * Hs * Ws * Cout * Cin * 16

                sums = torch.empty(int(lib.ms_subpix_pack_floats(Cin, Cout)), device=dev)
                check(lib.ms_subpix_pack(wp.data_ptr(), sums.data_ptr(), Cin, Cout, st), "ms_subpix_pack")

                def run(flags=0):
                    check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), sums.data_ptr(), 0, N, Cin, Hs, Ws, Cout, 0, stats.data_ptr(), 0, 0, 0, 1.0, 0, flags, st), "ms_conv_subpix2")
            else:
                # data-gradient of a stride-2 conv with Cin_fwd = Cout (of this call), Cout_fwd = Cin (of this call)
                w = (torch.randn(Cin, Cout, 3, 3, generator=g) * (1.0 / (3.0 * Cin ** 0.5))).to(dev)
                wp = ops.pack_conv_weight_dgrad(w)
                flop = 2.0 * N * Hs * Ws * Cout * Cin * 9
                if kind == "dgm":
                    u = torch.randn(N, Cout, Ho, Ho, generator=g).to(dev)
                    coef = torch.randn(Cout, 4, generator=g).to(dev)
                    tab = torch.zeros(int(lib.ms_conv_actbwd_tab_bytes(Cout)) // 4, device=dev)

                    def run(flags=0):
                        check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), 0, 0, N, Cin, Hs, Ws, Cout, 1, 0, 0, u.data_ptr(), coef.data_ptr(), 0.2, tab.data_ptr(), flags, st), "ms_conv_subpix2")
                else:
                    def run(flags=0):
                        check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), 0, 0, N, Cin, Hs, Ws, Cout, 1, 0, 0, 0, 0, 1.0, 0, flags, st), "ms_conv_subpix2")
        variants = [("gen1", -1), ("gen2", -2), ("gen1", -1), ("gen2", -2)] if kind == "s2" else [("gen1", 1), ("tiles", 2), ("blocks", 4), ("auto", 0)]
        line = f"{idx:2d} {kind:4s} {N}x{Cin}x{Hs}x{Ws} -> {Cout:4d} "
        sums0 = None
        for vname, flags in variants:
            if flags < 0:                      # the stride-2 forward: a process-wide switch, not a per-call flag
                lib.ms_set_option(b"conv.s2g2", 1 if flags == -2 else 0)
            out.zero_()
            for _ in range(3):
                run(flags)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(flags); e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            us = ts[len(ts) // 2]
            cs = bits_sum(out)
            same = "" if sums0 is None else (" =" if cs == sums0 else " DIFFERENT-BITS")
            if sums0 is None:
                sums0 = cs
            line += f" | {vname} {us:7.1f} us {flop / PEAK / (us * 1e-6):.3f}{same}"
        lib.ms_set_option(b"conv.s2g2", 1)
        print(line + f"   out {sums0:012x}", flush=True)


if __name__ == "__main__":
    main()
