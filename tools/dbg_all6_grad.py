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
# configurations: engine options (EngineOptions fields) and library options ("a.b" names), maxstyle_amd/options.py
from maxstyle_amd import options as O
CONFIGS = [dict(winograd=True), dict(winograd=False), dict(winograd=True, pool_epi=False), dict(winograd=True, pool_epi=False, pool_fuse=False),
           dict(winograd=True, fuse_act_bwd=False), dict(winograd=True, ride=False), dict(winograd=True, xfin_pro=False),
           {"winograd": True, "conv.k1s": 0, "conv.k1g": 0, "conv.s2g2": 0}]
if len(sys.argv) > 1:
    CONFIGS = [{k: int(v) for k, v in (kv.split("=") for kv in a.split(","))} for a in sys.argv[1:]]
for cfg in CONFIGS:
    libopts = {k: v for k, v in cfg.items() if "." in k}
    O._engine_defaults.clear(); O._engine_defaults.update({k: bool(v) for k, v in cfg.items() if "." not in k})
    prev = {k: O.set_library_option(k, v) for k, v in libopts.items()}
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
    for k, v in prev.items():
        O.set_library_option(k, v)
