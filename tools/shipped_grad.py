"""Shipped workloads: the step-1 gradient of every style tensor against the reference's fp64 gradient at the same point (tests/golden/loop_shipped_*.npz), next to the
reference's own fp32 gradient error - separates "the loop is chaotic" from "a kernel is off" at the 12- / 14- / 24- / 28-pixel levels (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import r5_cases as R5
from maxstyle_amd.options import engine_defaults
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["acdc", "prostate"]
for w in which:
    for wino in (True, False):
        with engine_defaults(winograd=wino):
            g, spec, img, lab, styles, layers = R5.shipped_inputs(w, dev)
            S = R5.shipped_solver(dev, w)
            def hook(mods):
                for k, m in mods.items():
                    st = styles[int(k)]
                    m.perm = st.perm.clone()
                    with torch.no_grad():
                        m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev); m.lmda.data = st.lmda.to(dev)
            S.style_init_hook = hook
            z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
            S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=1, lr=0.1, always_use_beta=bool(R5.SHIPPED[w]["beta"]),
                                       reference_image=img.to(dev), reference_segmentation=lab.to(dev))
            eng = next(iter(S._engines.values()))
        print(f"== {w} {'winograd' if wino else 'direct'}: step-1 gradient max|err| / max|g|: ours | reference fp32      loss rel err {abs(float(S.last_losses[0]) - g['f64.losses'][0]) / abs(g['f64.losses'][0]):.1e}")
        for i in layers:
            for nm in ("gamma_noise", "beta_noise", "lmda"):
                ours = eng.grad(i, nm).detach().cpu().numpy().astype(np.float64).reshape(-1)
                r64 = g[f"f64.step1.grad.{i}.{nm}"].reshape(-1); r32 = g[f"f32.step1.grad.{i}.{nm}"].astype(np.float64).reshape(-1)
                m = np.abs(r64).max()
                eo, er = np.abs(ours - r64), np.abs(r32 - r64)
                print(f"   {i}.{nm:12s} max|g| {m:.2e}   ours {eo.max() / m:.2e} | ref {er.max() / m:.2e}    l2: ours {np.linalg.norm(ours - r64) / np.linalg.norm(r64):.2e} | ref {np.linalg.norm(r32 - r64) / np.linalg.norm(r64):.2e}   worst sample/channel {int(eo.argmax())}")
