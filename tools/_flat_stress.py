"""Flat vs tiled Winograd form at production shapes: alternating inputs (stale-LDS reads would show), outputs bit for bit, tables to rounding (scratch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops
from maxstyle_amd.options import library_option
dev = torch.device("cuda:0")
def _rand(shape, seed, s=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g) * s
U = ops.FETCH_WINOGRAD | ops.FETCH_WINO_U
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (N, Cin, Cout, H, W) in [(20, 128, 128, 28, 28), (16, 512, 512, 20, 20), (20, 64, 128, 28, 28), (20, 128, 128, 24, 24)]:
    ins = []
    for s in (0, 100):
        x = _rand((N, Cin, H, W), 1 + s).to(dev); x2 = _rand((N, Cin, H, W), 2 + s).to(dev)
        cfd = _rand((Cin, 4), 5 + s).to(dev)
        u = (_rand((N, Cout, H, W), 24 + s) + 0.3).to(dev)
        ins.append((x, x2, cfd, u))
    w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4).to(dev)
    wp, has = ops.with_wino_appendix(ops.pack_conv_weight(w.to(dev)), Cin, Cout)
    base = _rand((N, Cout, H, W), 6).to(dev)
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1).to(dev)
    one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    def run(i):
        x, x2, cfd, u = ins[i]
        pa, pb, pc = ops.coef_ptrs(cfd)[:3]
        r = {}
        st, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
        r["fwd"] = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U, stats=st)
        r["fwd.coef"] = ops.bn_finalize(st, parts, one, zero)
        st1, _ = ops.conv_stats_buffer(N, Cout, H, W, dev)
        r["pro1"] = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U, stats=st1, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
        r["pro1.coef"] = ops.bn_finalize(st1, parts, one, zero)
        kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2)
        r["pro2"] = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, **kw)
        r["acc"] = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, out=base.clone(), epi_mode=1, **kw)
        g, t = ops.conv2d_actbwd(x, wp, Cout, 3, u, coef4, 0.2, fetch=U, **kw)
        r["actbwd"] = g; r["actbwd.coef"] = ops.bn_bwd_coefs(t, 0, coef4, N * H * W)
        return r
    with library_option("conv.wino_flat", 0):
        tiled = [run(0), run(1)]
        torch.cuda.synchronize()
    bad = {}
    with library_option("conv.wino_flat", 2):
        assert ops.lib.ms_conv2d_form(N, Cin, H, W, Cout, 0, 0, U) == 7
        for it in range(REP):
            fs = [run(it & 1), run((it & 1) ^ 1)]
            torch.cuda.synchronize()
            for q, f in enumerate(fs):
                t = tiled[(it & 1) ^ q]
                for k in t:
                    if k.endswith(".coef"):
                        e = float((f[k] - t[k]).abs().max()) / float(t[k].abs().max())
                        if e > 1e-5: bad.setdefault(k, []).append((it, q, e))
                    elif not torch.equal(f[k], t[k]):
                        d = (f[k] != t[k]).nonzero()
                        bad.setdefault(k, []).append((it, q, int(d.shape[0]), d[:2].tolist(), d[-1].tolist(), float((f[k] - t[k]).abs().max())))
    print((N, Cin, Cout, H, W), "OK" if not bad else "", flush=True)
    for k, v in bad.items():
        print("  ", k, len(v), "of", 2 * REP, "runs differ; first:", v[:3], flush=True)
