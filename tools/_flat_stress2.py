"""Flat form run to run: statistics tables bit for bit (scratch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import ops
from maxstyle_amd.options import library_option
dev = torch.device("cuda:0")
def _rand(shape, seed, s=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g) * s
U = ops.FETCH_WINOGRAD | ops.FETCH_WINO_U
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for flat in (2, 0):
  for (N, Cin, Cout, H, W) in [(16, 512, 512, 20, 20), (20, 128, 128, 28, 28)]:
    x = _rand((N, Cin, H, W), 1).to(dev)
    cfd = _rand((Cin, 4), 5).to(dev); pa, pb, pc = ops.coef_ptrs(cfd)[:3]
    w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4).to(dev)
    wp, has = ops.with_wino_appendix(ops.pack_conv_weight(w.to(dev)), Cin, Cout)
    one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    def run():
        r = {}
        st, parts = ops.conv_stats_buffer(N, Cout, H, W, dev); st.zero_()
        r["fwd"] = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U, stats=st)
        r["fwd.stats"] = st
        r["fwd.coef"] = ops.bn_finalize(st, parts, one, zero)
        st1, _ = ops.conv_stats_buffer(N, Cout, H, W, dev); st1.zero_()
        r["pro1"] = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U, stats=st1, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
        r["pro1.stats"] = st1
        r["pro1.coef"] = ops.bn_finalize(st1, parts, one, zero)
        return r
    with library_option("conv.wino_flat", flat):
        ref = run(); torch.cuda.synchronize()
        nbad = 0
        for it in range(REP):
            f = run(); torch.cuda.synchronize()
            for k in ref:
                if not torch.equal(f[k], ref[k]):
                    d = (f[k] != ref[k]).nonzero()
                    nbad += 1
                    if nbad <= 6:
                        print("   flat", flat, (N, Cin, Cout, H, W), "run", it, k, "differs at", int(d.shape[0]), "elements; first", d[:6].tolist(), "values", f[k][tuple(d[0].tolist())].item(), ref[k][tuple(d[0].tolist())].item(), flush=True)
        print("flat", flat, (N, Cin, Cout, H, W), "differing (run, tensor) pairs:", nbad, "of", REP * len(ref), flush=True)
