"""Timing-only ablations of one conv form (MS_CONV_DBG bits: 1 no MFMA loop, 2 no global loads, 8 no LDS stores, 4 no epilogue stores) at 16x16x256x256 and 16x64x64x64."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    from maxstyle_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for (N, C, H) in ((16, 16, 256), (16, 64, 64)):
        x = torch.randn(N, C, H, H, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.1
        wp = ops.pack_conv_weight(w)
        y = torch.empty_like(x)
        stats, parts = ops.conv_stats_buffer(N, C, H, H, dev)
        f = lambda: ops.conv2d(x, wp, None, C, 3, 1, stats=stats, out=y)
        for _ in range(3): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        out[f"{C}ch@{H}"] = e0.elapsed_time(e1) * 50
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for form, env in (("x3", {"MS_CONV_X3": "2", "MS_CONV_WINO": "0"}), ("winograd", {"MS_CONV_X3": "0", "MS_CONV_WINO": "2"}), ("direct", {"MS_CONV_X3": "0", "MS_CONV_WINO": "0"})):
            for dbg in (0, 1, 8, 2, 4, 16):
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, MS_CONV_DBG=str(dbg), **env), capture_output=True, text=True, timeout=600)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                print(form, "dbg", dbg, line[0] if line else p.stderr[-300:])
