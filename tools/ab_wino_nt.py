"""A/B of the Winograd form's channel blocks per staged tile (round 4): every distinct Winograd-eligible 3x3 call of one inner step with more than 16 output
channels, replayed on its live buffers with MS_FETCH_WINO_NT1 (one block per tile: the round-3 kernel) and without (two blocks: ms_conv_inst_wino2.hip).
Run on the GPU box:  python tools/ab_wino_nt.py [c2|c4|acdc192|prostate224] [reps]      (round 6: the reference's shipped shapes, batch 20; a one-block block-form column)
The engine is built with the options xfin = ride = False so that every convolution goes through ms_conv2d / ms_conv2d_actbwd (the `_xfin` twins launch the same kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maxstyle_amd import options as _O
_O._engine_defaults.update(xfin=False, ride=False)
_O.set_library_option("conv.wino_nt", 2)      # NT1 bit decides per call; "auto" column: the heuristic (option back to 0)
import torch
import bench
from maxstyle_amd import _lib

NT1 = 0x400


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    net, size, B = {"c2": ((4, 1, 4), 256, 16), "c4": ((1, 3, 2), 320, 16), "acdc192": ((4, 1, 4), 192, 20), "prostate224": ((4, 1, 2), 224, 20)}[cfg]
    eng, W, img, lab, styles, z_i, lab_d = bench.build(dev, B, size, 0, net)
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
    seen = {}
    for name, a in calls:
        fi = 12 if name == "ms_conv2d" else 11
        if name == "ms_conv2d":
            N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[5:14]
            epi = a[20]; stats = a[21] != 0
        else:
            N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[4:13]
            epi, stats = 3, False
        if not (ks == 3 and stride == 1 and (fetch & 0x100) and (fetch & 0xFF) == 0 and Cout > 16 and Ws >= 20 and Cin % 8 == 0):      # (Cout > 16: the one-block-only layers have nothing to choose)
            continue
        key = (name, N, Cin, Hs, Ws, Cout, pm, epi, stats)
        seen.setdefault(key, [0, a, fi])[0] += 1
    st = torch.cuda.current_stream()

    def time_call(name, a):
        fn = orig[name]
        assert fn(*a) == 0
        for _ in range(3):
            fn(*a)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn(*a)
            e1.record(st)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    # columns: one / two channel blocks per staged tile, weights transformed in the kernel (the MS_FETCH_WINO_U bit cleared) or staged from the appendix (+U)
    WU = 0x800
    BLK = 0x1000
    print(f"{'call':12s} {'N,Cin,Hs,Ws,Cout':>24s} pm epi st cnt   nt1_us  nt1+U_us   nt2_us  nt2+U_us  blk2+U_us  blk1+U_us  exec_mfma_frac(nt1 -> best)   auto")
    tot = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    for key, (cnt, a, fi) in sorted(seen.items(), key=lambda kv: -kv[1][0] * kv[0][2] * kv[0][5] * kv[0][3] * kv[0][4]):
        name, N, Cin, Hs, Ws, Cout, pm, epi, stats = key
        has_u = bool(a[fi] & WU)
        ts = []
        _O.set_library_option("conv.wino_nt", 2)         # two blocks wherever the call's own NT1 bit does not say one
        _O.set_library_option("conv.wino_block", 0)      # tiled columns: never the block form
        for nt1, wu in ((1, 0), (1, 1), (0, 0), (0, 1)):
            a1 = list(a); a1[fi] = (a[fi] & ~WU & ~NT1) | (NT1 if nt1 else 0) | (WU if (wu and has_u) else 0)
            ts.append(time_call(name, tuple(a1)))
        a1 = list(a); a1[fi] = (a[fi] & ~NT1) | BLK
        ts.append(time_call(name, tuple(a1)))        # the block form, two channel blocks, weights from the appendix when the engine packed one
        a1 = list(a); a1[fi] = a[fi] | NT1 | BLK
        ts.append(time_call(name, tuple(a1)))        # the block form, one channel block
        _O.set_library_option("conv.wino_block", 1)
        _O.set_library_option("conv.wino_nt", 0)         # 0 = the heuristic
        _O.set_library_option("conv.wino_flat", 2)       # (round 6) the flat form wherever legal ...
        flat = time_call(name, a)
        _O.set_library_option("conv.wino_flat", 0)       # ... never ...
        noflat = time_call(name, a)
        _O.set_library_option("conv.wino_flat", 1)       # ... by its rounds rule (the default)
        auto = time_call(name, a)
        for i, t in enumerate(ts + [auto]):
            tot[i] += cnt * t
        # executed matrix work: 16 MFMAs of 16x16x4 per (2x2 tile group of 16, 4 channels, 16 output channels): 16/36 of the direct form's flops
        ex = 2.0 * N * Hs * Ws * Cout * Cin * 9 * 16 / 36 / 157.3e12 * 1e6
        print(f"{name[3:]:12s} {str((N, Cin, Hs, Ws, Cout)):>24s} {pm:2d} {epi:3d} {int(stats):2d} {cnt:3d} {ts[0]:8.1f} {ts[1]:8.1f} {ts[2]:8.1f} {ts[3]:8.1f} {ts[4]:8.1f} {ts[5]:8.1f}     {ex / ts[0]:5.2f} -> {ex / min(ts):5.2f}    {auto:8.1f}   flat forced / off {flat:7.1f} / {noflat:7.1f}")
    print(f"per step: nt1 {tot[0]:.0f} us, nt1+U {tot[1]:.0f} us, nt2 {tot[2]:.0f} us, nt2+U {tot[3]:.0f} us, blocks {tot[4]:.0f} us, one-block blocks {tot[5]:.0f} us; the dispatch's own choice {tot[6]:.0f} us")


if __name__ == "__main__":
    main()
