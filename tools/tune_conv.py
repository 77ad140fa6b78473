"""Per-layer convolution tuning table of one inner step (run on the GPU box).

Records every ms_conv2d / ms_conv2d_actbwd call of one eager step at the C2 configuration, then replays each distinct call (same live buffers)
under the library's tuning options (ms_set_option "conv.force_nt": output-channel tile NT forced to 1/2/4; "conv.wide": wide-read kernel on/off) and prints what the built-in
heuristic chose against the best alternative.  Usage: python tools/tune_conv.py [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maxstyle_amd import options as _O
_O._engine_defaults.update(xfin=False, ride=False)      # every conv through ms_conv2d / ms_conv2d_actbwd (the `_xfin` / rider twins launch the same kernels)
import torch
import bench
from maxstyle_amd import _lib


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    c4 = "c4" in sys.argv[2:]                      # python tools/tune_conv.py 10 c4: BASELINE config 4's shapes (FCN_64, 16x3x320x320)
    eng, W, img, lab, styles, z_i, lab_d = bench.build(dev, 16, 320, 0, (1, 3, 2)) if c4 else bench.build(dev, 16, 256, 0, (4, 1, 4))
    eng.code, eng.labels = z_i, lab_d
    eng._prefix_valid = False
    im = eng.decode(z_i)
    eng.step(im)                                   # allocate everything
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
    seen = {}
    for name, a in calls:
        if name == "ms_conv2d":
            N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[5:14]
            epi = a[20]; stats = a[21] != 0
            key = (name, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm, epi, stats)
        else:
            N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[4:13]
            key = (name, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm, 3, False)
        seen.setdefault(key, [0, a])[0] += 1
    st = torch.cuda.current_stream()

    def time_call(name, a):
        fn = orig[name]
        rc = fn(*a)
        if rc != 0:
            return None
        for _ in range(3):
            fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn(*a)
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    rows = []
    total_auto = total_best = 0.0
    for key, (cnt, a) in seen.items():
        name, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm, epi, stats = key
        res = {}
        for nt in (0, 1, 2, 4):
            for wide in (1, 0):
                _O.set_library_option("conv.force_nt", nt); _O.set_library_option("conv.wide", wide)
                if nt == 0 and wide == 0:
                    continue
                t = time_call(name, a)
                if t is not None:
                    res[(nt, wide)] = t
        _O.set_library_option("conv.force_nt", 0); _O.set_library_option("conv.wide", 1)
        auto = res[(0, 1)]
        best = min(res, key=res.get)
        Ho = Hs * (2 if (fetch & 0xFF) else 1) // stride if ks != 2 else Hs // 2
        Wo = Ws * (2 if (fetch & 0xFF) else 1) // stride if ks != 2 else Ws // 2
        cols = 4 * Cout if epi == 2 else Cout
        flop = 2.0 * N * Ho * Wo * cols * Cin * ks * ks
        rows.append((cnt * auto, key, cnt, auto, best, res[best], flop / auto / 1e6))
        total_auto += cnt * auto; total_best += cnt * res[best]
    rows.sort(reverse=True)
    print(f"{'call':16s} {'N,Cin,Hs,Ws,Cout':>22s} ks s f  pm epi st  cnt  auto_us  TF/s   best(nt,wide) best_us  gain_us/step")
    for tot, key, cnt, auto, best, tb, tf in rows:
        name, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm, epi, stats = key
        print(f"{name[3:]:16s} {str((N, Cin, Hs, Ws, Cout)):>22s} {ks:2d} {stride} {str(fetch & 0xFF) + ('w' if fetch & 0x100 else ' ')} {pm:2d} {epi:3d} {int(stats):2d} {cnt:4d} {auto:8.1f} {tf:6.1f}   {str(best):>10s} {tb:8.1f} {cnt * (auto - tb):8.1f}")
    print(f"conv time per step: heuristic {total_auto:.0f} us, best-of-table {total_best:.0f} us")


if __name__ == "__main__":
    main()
