"""Replay ONE convolution launch of the C2 inner step on its live buffers, N times back to back (for rocprofv3 --pmc / --kernel-trace).

    python tools/replay_conv.py <which> [reps] [timing-only ablation bits of the conv kernels: diag.conv_dbg, results WRONG]
which: dgrad_actbwd | dgrad_plain | dgrad_acc | dgrad_nt2 | fwd_pro1 | fwd_pro0        (all at their dominant shape of the step)
The launch sequence of one eager step is recorded exactly as tools/tune_conv.py does; the selected call is then re-issued with the same arguments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from maxstyle_amd import _lib

# (entry point, Cin, Hs, pro_mode, epi_mode, stats)
WHICH = {
    "dgrad_actbwd": ("ms_conv2d_actbwd", 16, 256, 2, 3, False),     # conv_wide_kernel<1,2>, activation-backward epilogue (3 per step)
    "dgrad_plain": ("ms_conv2d", 16, 256, 2, 0, False),             # conv_wide_kernel<1,2>, plain epilogue
    "dgrad_acc": ("ms_conv2d", 16, 256, 2, 1, False),               # conv_wide_kernel<1,2>, accumulate epilogue
    "dgrad_nt2": ("ms_conv2d_actbwd", 32, 128, 2, 3, False),        # conv_wide_kernel<2,2>
    "fwd_pro1": ("ms_conv2d", 16, 256, 1, 0, True),                 # conv_wide_kernel<1,1> + statistics
    "fwd_pro0": ("ms_conv2d", 16, 256, 0, 0, True),                 # conv_wide_kernel<1,0> + statistics
}


def record_step(dev):
    eng, W, img, lab, styles, z_i, lab_d = bench.build(dev, 16, 256, 0, (4, 1, 4))
    eng.code, eng.labels = z_i, lab_d
    eng._prefix_valid = False
    im = eng.decode(z_i)
    eng.step(im)
    calls = []
    lib = _lib.lib
    orig = {n: getattr(lib, n) for n in ("ms_conv2d", "ms_conv2d_actbwd")}

    class Rec:
        def __init__(self, name):
            self.name = name

        def __call__(self, *a):
            calls.append((self.name, a))
            return orig[self.name](*a)
    import maxstyle_amd.engine as E, maxstyle_amd.ops as O

    class LibProxy:
        def __getattr__(self, n):
            return Rec(n) if n in orig else getattr(lib, n)
    E.lib = LibProxy(); O.lib = LibProxy()
    eng.step(im)
    E.lib = lib; O.lib = lib
    torch.cuda.synchronize()
    return eng, calls, orig


def describe(name, a):
    if name == "ms_conv2d":
        N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[5:14]
        return dict(N=N, Cin=Cin, Hs=Hs, Ws=Ws, Cout=Cout, ks=ks, stride=stride, fetch=fetch, pm=pm, epi=a[20], stats=a[21] != 0)
    N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[4:13]
    return dict(N=N, Cin=Cin, Hs=Hs, Ws=Ws, Cout=Cout, ks=ks, stride=stride, fetch=fetch, pm=pm, epi=3, stats=False)


def main():
    which = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dbg = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda:0")
    eng, calls, orig = record_step(dev)
    ep, cin, hs, pm, epi, stats = WHICH[which]
    for name, a in calls:
        d = describe(name, a)
        if name == ep and d["Cin"] == cin and d["Hs"] == hs and d["pm"] == pm and d["epi"] == epi and d["stats"] == stats and d["ks"] == 3 and d["stride"] == 1 and (d["fetch"] & 0xFF) == 0 \
                and d["Cout"] == cin:
            if dbg:
                from maxstyle_amd.options import set_library_option
                set_library_option("diag.conv_dbg", dbg)
            for _ in range(reps):
                rc = orig[name](*a)
                assert rc == 0
            torch.cuda.synchronize()
            print(which, d, "replayed", reps)
            return
    raise SystemExit(f"no call matches {which}")


if __name__ == "__main__":
    main()
