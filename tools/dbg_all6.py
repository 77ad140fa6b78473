"""all-six-layers case on the trained network: per-step loss errors and per-tensor parameter errors against the reference's fp64 run (GPU box)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import r3_cases as R
dev = torch.device("cuda:0")
r = R.arg_case(dev, "all6")
print("losses_rel", r["losses_rel"]); print("noise     ", r["noise_losses_rel"])
for k in r["params_rel"]:
    print(f"  {k:16s} {r['params_rel'][k]:.2e}  (reference fp32 {r['noise_params_rel'][k]:.2e})")
print("image", r["image_rel"], r["noise_image_rel"])
g = np.load(os.path.join(ROOT, "tests", "golden", "loop_args_all6.npz"))
S = R.trained_solver(dev, "trained_fcn16.npz")
mods = None
# step-1 parameters against the reference's (Adam's first step is lr * sign(g): an element off by 0.2 = a gradient whose sign differs)
r = R.arg_case(dev, "all6", S)
mods = S.last_style_modules
names = [str(n) for n in g["all6.param_names"]]
for n in names:
    i, nm = n.split(".")
    cur = getattr(mods[i], nm).detach().cpu().numpy().astype(np.float64).reshape(-1)
    r64 = g[f"all6.f64.final.{i}.{nm}"].astype(np.float64).reshape(-1); r32 = g[f"all6.f32.final.{i}.{nm}"].astype(np.float64).reshape(-1)
    d = np.abs(cur - r64); d32 = np.abs(r32 - r64)
    j = int(d.argmax())
    print(f"  final {n:16s} max abs diff {d.max():.3e} at {j} (ours {cur[j]:+.5f} ref64 {r64[j]:+.5f} ref32 {r32[j]:+.5f}); reference fp32 max {d32.max():.3e}; elements > 1e-3: {(d > 1e-3).sum()} of {d.size}")
for s in (1, 2, 3):
    for n in names:
        k64, k32 = f"all6.f64.step{s}.grad.{n}", f"all6.f32.step{s}.grad.{n}"
        if k64 in g.files:
            a, b = g[k64].astype(np.float64).reshape(-1), g[k32].astype(np.float64).reshape(-1)
            print(f"  step{s} grad {n:16s} |g|max {np.abs(a).max():.2e} min|g| {np.abs(a).min():.2e} ref32-vs-64 max {np.abs(a-b).max():.2e}  sign flips {(np.sign(a) != np.sign(b)).sum()}")
