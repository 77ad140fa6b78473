"""The three-way bf16 split mode of the wide conv kernel (MS_FETCH_X3 / MS_CONV_X3=2) against fp64, beside the direct and the Winograd fp32 forms; with timings.
    python tools/x3_check.py            (spawns itself once per form: the library reads MS_CONV_X3 / MS_CONV_WINO once)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64), (1, 8, 33, 10, 100), (2, 16, 16, 6, 72), (16, 16, 16, 256, 256), (16, 32, 32, 128, 128), (16, 64, 64, 64, 64), (4, 64, 64, 320, 320)]


def worker():
    import torch, torch.nn.functional as F
    from maxstyle_amd import ops
    dev = torch.device("cuda:0")
    rnd = lambda shape, seed, scale=1.0: torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale
    err = lambda o, r: float((o.cpu().double() - r).abs().max() / r.abs().max())
    res = {}
    for (N, Cin, Cout, H, W) in SHAPES:
        big = N * Cin * H * W > 4e6
        x = rnd((N, Cin, H, W), 1) * 0.7 + 0.5; x2 = rnd((N, Cin, H, W), 2); w = rnd((Cout, Cin, 3, 3), 3, 0.1); b = rnd((Cout,), 4)
        cf = rnd((Cin, 4), 5); cfd = cf.to(dev)
        wp = ops.pack_conv_weight(w.to(dev))
        xd, x2d = x.to(dev), x2.to(dev)
        stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
        out = torch.empty(N, Cout, H, W, device=dev)
        f = lambda: ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, stats=stats, out=out)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); e1.synchronize()
        r = {"us_fwd_stats": e0.elapsed_time(e1) * 100}
        if not big:
            a, bb, cc = (cf[:, i].double().view(1, -1, 1, 1) for i in range(3))
            ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
            r["plain"] = err(out, ref)
            coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
            r["mean"] = float((coef[:, 2] - ref.mean((0, 2, 3))).abs().max())
            o1 = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
            r["pro1"] = err(o1, F.conv2d(F.leaky_relu(a * x.double() + bb, 0.2), w.double(), None, padding=1))
            base = rnd((N, Cout, H, W), 6)
            o2 = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                            pro_cstride=4, in2=x2d, epi_mode=1, out=base.to(dev).clone())
            r["pro2_acc"] = err(o2, F.conv2d(a * x.double() + bb * x2.double() + cc, w.double(), None, padding=1) + base.double())
        else:
            pa, pb, pc = ops.coef_ptrs(cfd)
            g = lambda: ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2d, out=out)
            g(); torch.cuda.synchronize()
            e0.record()
            for _ in range(10): g()
            e1.record(); e1.synchronize()
            r["us_dgrad_pro2"] = e0.elapsed_time(e1) * 100
        res[str((N, Cin, Cout, H, W))] = r
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        rows = {}
        for tag, env in (("direct", {"MS_CONV_X3": "0", "MS_CONV_WINO": "0"}), ("winograd", {"MS_CONV_X3": "0", "MS_CONV_WINO": "2"}), ("x3", {"MS_CONV_X3": "2", "MS_CONV_WINO": "0"})):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", p.stderr[-1500:]); continue
            rows[tag] = json.loads(line[0])
        for shape in rows.get("direct", {}):
            print(shape)
            for tag in rows:
                r = rows[tag][shape]
                print("   %-9s " % tag + "  ".join(f"{k} {v:.3g}" for k, v in r.items()))
