"""all-six-layers case on the trained network: the step-1 gradient of every tensor against the reference's fp64 gradient, Winograd and direct form (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import r3_cases as R
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "loop_args_all6.npz"))
names = [str(n) for n in g["all6.param_names"]]
R.ARG_CALLS["all6"] = dict(n_iter=1)
CONFIGS = [dict(MS_LOOP_WINOGRAD="1"), dict(MS_LOOP_WINOGRAD="0"), dict(MS_LOOP_WINOGRAD="1", MS_POOL_EPI="0"), dict(MS_LOOP_WINOGRAD="1", MS_WINO_APPENDIX="0"),
           dict(MS_LOOP_WINOGRAD="1", MS_POOL_EPI="0", MS_POOL_FUSE="0"), dict(MS_LOOP_WINOGRAD="1", MS_FUSE_ACTBWD="0"), dict(MS_LOOP_WINOGRAD="1", MS_RIDE="0"),
           dict(MS_LOOP_WINOGRAD="1", MS_XFIN_PRO="0"), dict(MS_LOOP_WINOGRAD="1", MS_CONV_K1S="0", MS_CONV_K1G="0", MS_CONV_S2G2="0", MS_SUBPIX_GEN="1")]
if len(sys.argv) > 1:
    CONFIGS = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
BASE = dict(os.environ)
for cfg in CONFIGS:
    os.environ.clear(); os.environ.update(BASE); os.environ.update(cfg)
    wino = str(cfg)
    worst = 0.0
    S = R.trained_solver(dev, "trained_fcn16.npz")
    try:
        R.arg_case(dev, "all6", S)
    except Exception as e:
        print("arg_case raised (expected: fixture has 3 steps)", type(e).__name__, str(e)[:100])
    eng = next(iter(S._engines.values()))
    print(f"== {wino}: step-1 gradient, max abs error / max |g| (ours | reference fp32), and over the elements with |g| < 1e-3 max|g|: worst relative error")
    for n in names:
        i, nm = n.split(".")
        ours = eng.grad(int(i), nm).detach().cpu().numpy().astype(np.float64).reshape(-1)
        r64 = g[f"all6.f64.step1.grad.{n}"].astype(np.float64).reshape(-1); r32 = g[f"all6.f32.step1.grad.{n}"].astype(np.float64).reshape(-1)
        m = np.abs(r64).max()
        small = np.abs(r64) < 1e-3 * m
        rel_o = np.abs(ours - r64) / np.maximum(np.abs(r64), 1e-30); rel_r = np.abs(r32 - r64) / np.maximum(np.abs(r64), 1e-30)
        worst = max(worst, np.abs(ours - r64).max() / m)
        if os.environ.get("DBG_VERBOSE", "0") == "1": print(f"   {n:16s} max|g| {m:.2e}  ours {np.abs(ours - r64).max() / m:.2e} | ref {np.abs(r32 - r64).max() / m:.2e}   median rel ours {np.median(rel_o):.1e} ref {np.median(rel_r):.1e}   "
              f"worst rel ours {rel_o.max():.1e} ref {rel_r.max():.1e}  ({small.sum()} small)")
    print(f"   worst max-norm error over the 18 tensors: {worst:.2e}")
