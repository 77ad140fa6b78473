"""Round-5 parity cases on the GPU (functions that MEASURE; tests/test_round5_gpu.py asserts on them, bench.py reports them).

The reference's SHIPPED workloads - the sizes a user of the reference actually runs (VERDICT r4 missing 2):
  config/ACDC/1500_epoch/MICCAI2022_MaxStyle.json:24-28,51,56-76   20x1x192x192, FCN_16, 4 classes, K = 5, layers [3,4,5], always_use_beta false
  config/Prostate/MICCAI2022_MaxStyle.json:20-24,42,47-66           20x1x224x224, FCN_16, 2 classes, K = 5, layers [3,4,5], always_use_beta TRUE
against runs of the reference's generate_max_style_image at exactly those calls (tests/golden/make_golden_r5.py shipped -> loop_shipped_*.npz).
Their decoder levels are 192/96/48/24/12 and 224/112/56/28/14 pixels: the 12- / 14- / 24- / 28-pixel rows are shapes no other fixture has."""
import os

import numpy as np
import torch

from parity_util import rel
from r3_cases import GOLDEN, PN, segment, dice, load_trained

SHIPPED = {
    "acdc": dict(fixture="loop_shipped_acdc.npz", weights="trained_fcn16_192.npz", spec=(4, 1, 4), size=192, beta=False),
    "prostate": dict(fixture="loop_shipped_prostate.npz", weights="trained_fcn16_p224.npz", spec=(4, 1, 2), size=224, beta=True),
}


def shipped_solver(dev, which):
    import maxstyle_amd as M
    c = SHIPPED[which]
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=c["spec"][1], num_classes=c["spec"][2], use_gpu=True)
    W = load_trained(c["weights"])
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)
        mod.train()
    return S


def shipped_inputs(which, dev):
    """(image, labels, injected style states) of the shipped call - the generator's own (make_golden_r5.shipped_case), restated on the product's synthetic module."""
    from maxstyle_amd import synthetic as syn
    c = SHIPPED[which]
    g = np.load(os.path.join(GOLDEN, c["fixture"]))
    spec = syn.NetSpec(*c["spec"])
    B, layers = int(g["B"]), [int(i) for i in g["layers"]]
    img, lab = syn.synthetic_batch(B, c["size"], spec.image_ch, spec.num_classes, seed=1234)
    styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
    for i in layers:
        styles[i].lmda = torch.from_numpy(g[f"initial.{i}.lmda"])           # Beta(0.1, 0.1) draws for the Prostate call; equal to random_style_state's for ACDC
    return g, spec, img, lab, styles, layers


def shipped_case(dev, which, options=None):
    """generate_max_style_image through the drop-in solver at the reference's shipped call, against the REFERENCE's own fp64 run of exactly that call;
    noise_* = the reference's own fp32 run against its fp64 run.  `options`: engine options of the loop (e.g. {"winograd": False})."""
    g, spec, img, lab, styles, layers = shipped_inputs(which, dev)
    c = SHIPPED[which]
    K, ncls = int(g["K"]), spec.num_classes
    S = shipped_solver(dev, which)
    if options:
        S.loop_options = dict(options)
    img_d, lab_d = img.to(dev), lab.to(dev)

    def hook(mods):
        for k, m in mods.items():
            st = styles[int(k)]
            m.perm = st.perm.clone()
            with torch.no_grad():
                m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev); m.lmda.data = st.lmda.to(dev)
    S.style_init_hook = hook
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    zf = z_i.detach().reshape(-1).cpu()
    idx = torch.linspace(0, zf.numel() - 1, 4096).long()
    out = S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=K, lr=0.1, mix_style=True, no_noise=False, mix_learnable=True, noise_learnable=True,
                                     loss_types=["seg"], loss_weights=[1], always_use_beta=bool(c["beta"]), reference_image=img_d, reference_segmentation=lab_d)
    eng = next(iter(S._engines.values()))
    losses = S.last_losses.cpu().numpy().astype(np.float64)
    ref = torch.from_numpy(g["f64.image"]).double()
    scale = float(g["image_scale"])
    d = out.cpu().double() - ref
    pred = segment(S, out).argmax(1).cpu()
    clean_pred = segment(S, img_d).argmax(1).cpu()
    dsty, dclean = dice(pred, lab, ncls), dice(clean_pred, lab, ncls)
    return {
        "winograd": bool(eng.winograd),
        "z_i_rel": rel(zf[idx], g["f64.z_i.sample"]),
        "image_max": float(d.abs().max()) / scale, "noise_image_max": float(g["ref_noise.image_max"]),
        "image_rms": float(d.pow(2).mean().sqrt()) / scale, "noise_image_rms": float(g["ref_noise.image_rms"]),
        "image_max_per_sample": (d.abs().amax(dim=(1, 2, 3)) / scale).tolist(), "image_rms_per_sample": (d.pow(2).mean(dim=(1, 2, 3)).sqrt() / scale).tolist(),
        "losses_rel": (np.abs(losses - g["f64.losses"]) / np.abs(g["f64.losses"])).tolist(), "noise_losses_rel": g["ref_noise.losses_rel"].tolist(),
        "losses": losses.tolist(),
        "labels_equal_f64": float((pred.numpy() == g["f64.final_pred"]).mean()), "noise_labels_equal": float(g["ref_noise.labels_equal"]),
        "clean_labels_equal": float((clean_pred.numpy() == g["f64.clean_pred"]).mean()),
        "dice": dsty, "dice_ref_f64": g["f64.final_dice"].tolist(), "dice_ref_f32": g["f32.final_dice"].tolist(),
        "dice_clean": dclean, "dice_clean_ref": g["f64.clean_dice"].tolist(),
        "dice_abs_diff": max(abs(a - b) for a, b in zip(dsty, g["f64.final_dice"])),
        # the reference's own fp32 evaluations of this call (tests/golden/make_golden_r5.py shipped_draws: oneDNN at 8 / 2 threads, ATen native), each against its fp64 run
        "draws": {k[len("ref_draws."):]: g[k].tolist() for k in g.files if k.startswith("ref_draws.") and g[k].dtype.kind == "f"},
        "std_rel": {f"{i}.{nm}": rel(eng.buf[f"st{i}.std"][j].reshape(-1), g[f"f64.{i}.{nm}"].reshape(-1)) for i in layers for j, nm in enumerate(("gamma_std", "beta_std"))},
        "params_rel": {f"{i}.{nm}": rel(getattr(S.last_style_modules[str(i)], nm), g[f"f64.step{K}.param.{i}.{nm}"]) for i in layers for nm in PN},
        "noise_params_rel": {f"{i}.{nm}": rel(g[f"f32.step{K}.param.{i}.{nm}"], g[f"f64.step{K}.param.{i}.{nm}"]) for i in layers for nm in PN},
    }


def shipped_step1_gradients(dev, which):
    """The step-1 gradient of every style tensor (one evaluation at the injected parameters: nothing chaotic yet) against the reference's fp64 gradient at the same point;
    `draws` = the same error for each of the reference's own fp32 evaluations, per tensor."""
    g, spec, img, lab, styles, layers = shipped_inputs(which, dev)
    S = shipped_solver(dev, which)

    def hook(mods):
        for k, m in mods.items():
            st = styles[int(k)]
            m.perm = st.perm.clone()
            with torch.no_grad():
                m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev); m.lmda.data = st.lmda.to(dev)
    S.style_init_hook = hook
    z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=1, lr=0.1, always_use_beta=bool(SHIPPED[which]["beta"]),
                               reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    eng = next(iter(S._engines.values()))
    names = [str(n) for n in g["ref_draws.tensor_names"]]
    ours = {}
    for n in names:
        i, nm = n.split(".")
        r64 = g[f"f64.step1.grad.{n}"].reshape(-1)
        ours[n] = float(np.abs(eng.grad(int(i), nm).detach().cpu().numpy().astype(np.float64).reshape(-1) - r64).max() / np.abs(r64).max())
    return {"winograd": bool(eng.winograd), "ours": ours, "draws": {n: g["ref_draws.step1_grad_err"][:, j].tolist() for j, n in enumerate(names)},
            "first_loss_rel": abs(float(S.last_losses[0]) - float(g["f64.losses"][0])) / abs(float(g["f64.losses"][0]))}


def shipped_teacher_forced(dev, which):
    """TEACHER-FORCED steps of the shipped call (fixture keys `forced.*`, tests/golden/make_golden_r5.py shipped_forced): for k = 1..K the style parameters are set to what the
    reference's fp64 run held BEFORE its step k, one step is evaluated, and its loss and the gradient of every style tensor are compared with the reference's fp64 evaluation at
    the same point.  Nothing chaotic: five points of the reference's own trajectory, a smooth map at each.  -> per step: loss error, per-tensor gradient error (max norm over
    max|g|), and the same errors of the reference's own fp32 evaluations at that point (`draws`)."""
    g, spec, img, lab, styles, layers = shipped_inputs(which, dev)
    K = int(g["K"])
    S = shipped_solver(dev, which)
    names = [str(n) for n in g["ref_draws.tensor_names"]]
    img_d, lab_d = img.to(dev), lab.to(dev)
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    steps = []
    for k in range(1, K + 1):
        def hook(mods, k=k):
            for key, m in mods.items():
                st = styles[int(key)]
                m.perm = st.perm.clone()
                with torch.no_grad():
                    for nm in PN:
                        v = getattr(st, nm) if k == 1 else torch.from_numpy(g[f"f64.step{k - 1}.param.{key}.{nm}"]).float()
                        getattr(m, nm).data = v.to(dev)
                if k > 1:      # the batch std frozen by the run's first forward is part of the state of step k (maxstyle.py:165-168)
                    m.gamma_std = torch.from_numpy(g[f"f64.{key}.gamma_std"]).float().reshape(1, -1, 1, 1).to(dev)
                    m.beta_std = torch.from_numpy(g[f"f64.{key}.beta_std"]).float().reshape(1, -1, 1, 1).to(dev)
        S.style_init_hook = hook
        S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=1, lr=0.1, always_use_beta=bool(SHIPPED[which]["beta"]),
                                   reference_image=img_d, reference_segmentation=lab_d)
        eng = next(iter(S._engines.values()))
        l64 = float(g["f64.losses"][k - 1])
        if k > 1:
            assert abs(float(g[f"forced.f64.step{k}.loss"]) - l64) <= 1e-9 * abs(l64)      # the forced evaluation IS the free-running fp64 run's step k
        ours, draws = {}, {}
        for j, n in enumerate(names):
            i, nm = n.split(".")
            r64 = (g[f"f64.step1.grad.{n}"] if k == 1 else g[f"forced.f64.step{k}.grad.{n}"]).reshape(-1)
            ours[n] = float(np.abs(eng.grad(int(i), nm).detach().cpu().numpy().astype(np.float64).reshape(-1) - r64).max() / np.abs(r64).max())
            draws[n] = (g["ref_draws.step1_grad_err"][:, j] if k == 1 else g["forced.draws.grad_err"][:, k - 2, j]).tolist()
        steps.append({"k": k, "loss_rel": abs(float(S.last_losses[0]) - l64) / abs(l64),
                      "draw_loss_rel": (g["ref_draws.losses_rel"][:, 0] if k == 1 else g["forced.draws.loss_rel"][:, k - 2]).tolist(), "ours": ours, "draws": draws})
    return {"winograd": bool(eng.winograd), "steps": steps}


def full_teacher_forced(dev, which):
    """shipped_teacher_forced's method at the BENCHMARKED calls (`which` = "c2": trained FCN_16 at 16x1x256x256, K = 5, loop_full_c2.npz | "c4": trained FCN_64 at
    16x3x320x320, K = 10, loop_full_c4.npz): for k = 1..K one step from the parameters and frozen batch std the reference's fp64 run held before its step k, against the
    reference's fp64 evaluation at that point and next to its own fp32 evaluations there (tests/golden/loop_forced_full.npz, make_golden_r5.py forced_full)."""
    from maxstyle_amd import synthetic as syn
    import r3_cases as R3
    import r4_cases as R4
    g = np.load(os.path.join(GOLDEN, "loop_full_c2.npz" if which == "c2" else "loop_full_c4.npz"))
    f = np.load(os.path.join(GOLDEN, "loop_forced_full.npz"))
    if which == "c2":
        spec, size = syn.NetSpec(4, 1, 4), 256
        S = R3.trained_solver(dev, "trained_fcn16_256.npz")
    else:
        spec, size = syn.NetSpec(1, 3, 2), 320
        S = R4.trained_solver64(dev)
    B, layers, K = 16, [3, 4, 5], int(g["K"])
    assert int(f[f"{which}.steps_done"]) == K
    names = [str(n) for n in f[f"{which}.tensor_names"]]
    img, lab = syn.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
    styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
    img_d, lab_d = img.to(dev), lab.to(dev)
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    steps = []
    for k in range(1, K + 1):
        def hook(mods, k=k):
            for key, m in mods.items():
                st = styles[int(key)]
                m.perm = st.perm.clone()
                with torch.no_grad():
                    for nm in PN:
                        v = getattr(st, nm) if k == 1 else torch.from_numpy(g[f"f64.step{k - 1}.param.{key}.{nm}"]).float()
                        getattr(m, nm).data = v.to(dev)
                if k > 1:
                    m.gamma_std = torch.from_numpy(g[f"f64.{key}.gamma_std"]).float().reshape(1, -1, 1, 1).to(dev)
                    m.beta_std = torch.from_numpy(g[f"f64.{key}.beta_std"]).float().reshape(1, -1, 1, 1).to(dev)
        S.style_init_hook = hook
        S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=1, lr=0.1, reference_image=img_d, reference_segmentation=lab_d)
        eng = next(iter(S._engines.values()))
        l64 = float(f[f"{which}.step{k}.loss"])
        ours, draws = {}, {}
        for j, n in enumerate(names):
            i, nm = n.split(".")
            r64 = f[f"{which}.step{k}.grad.{n}"].reshape(-1)
            ours[n] = float(np.abs(eng.grad(int(i), nm).detach().cpu().numpy().astype(np.float64).reshape(-1) - r64).max() / np.abs(r64).max())
            draws[n] = f[f"{which}.draws.grad_err"][:, k - 1, j].tolist()
        steps.append({"k": k, "loss_rel": abs(float(S.last_losses[0]) - l64) / abs(l64), "draw_loss_rel": f[f"{which}.draws.loss_rel"][:, k - 1].tolist(), "ours": ours, "draws": draws})
    return {"winograd": bool(eng.winograd), "steps": steps}


# ------------------------------------------------------------------------------------------------------------- kink census at the benchmarked sizes
def one_step_buffers(dev, which, winograd):
    """ONE inner step (n_iter = 1) of the benchmarked call `which` ("c2": trained FCN_16 at 16x1x256x256 | "c4": trained FCN_64 at 16x3x320x320) with the given conv form
    -> the encoder / segmentor buffers of the engine (raw conv outputs and BatchNorm records: everything a LeakyReLU mask of the backward pass is decided from)."""
    from maxstyle_amd import synthetic as syn
    from maxstyle_amd.options import engine_defaults
    import r3_cases as R3
    import r4_cases as R4
    with engine_defaults(winograd=winograd):
        if which == "c2":
            spec, size = syn.NetSpec(4, 1, 4), 256
            S = R3.trained_solver(dev, "trained_fcn16_256.npz")
        else:
            spec, size = syn.NetSpec(1, 3, 2), 320
            S = R4.trained_solver64(dev)
        B, layers = 16, [3, 4, 5]
        img, lab = syn.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
        styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
        S.style_init_hook = R4._inject_all(styles, dev)
        z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
        S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=1, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
        eng = next(iter(S._engines.values()))
        assert eng.winograd == winograd
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in eng.buf.items() if torch.is_tensor(v) and v.is_floating_point() and (k.startswith("e.") or k.startswith("s."))}


def kink_census(bufs_a, bufs_b, eps=2e-5):
    """For every raw conv output u whose BatchNorm + LeakyReLU mask the backward pass uses (`*.u1` / `*.ua` with `*.bn1.coef`, `*.u` with `*.bn.coef`): the largest forward
    difference between the two runs (relative to the tensor's range), the elements whose masks differ, whether each of them has |pre-activation| < eps in BOTH runs, and the
    size of the kink set (elements within eps of zero in either run)."""
    out = {"tensors": 0, "elements": 0, "flips": 0, "flips_outside_kink_set": 0, "kink_set": 0, "worst_forward_rel": 0.0, "worst_flip_pre": 0.0, "per_tensor": {}}
    for k, a in bufs_a.items():
        if not (k.endswith(".u1") or k.endswith(".ua") or k.endswith(".u")):
            continue
        ck = k.rsplit(".", 1)[0] + (".bn.coef" if k.endswith(".u") else ".bn1.coef")
        if ck not in bufs_a or k not in bufs_b:
            continue
        b = bufs_b[k]
        fwd = float((a - b).abs().max()) / float(b.abs().max())
        pre = [c[:, 0].view(1, -1, 1, 1) * u + c[:, 1].view(1, -1, 1, 1) for u, c in ((a, bufs_a[ck]), (b, bufs_b[ck]))]
        diff = (pre[0] > 0) != (pre[1] > 0)
        nd = int(diff.sum())
        near = (pre[0].abs() < eps) | (pre[1].abs() < eps)
        outside = int((diff & ~((pre[0].abs() < eps) & (pre[1].abs() < eps))).sum())
        out["tensors"] += 1; out["elements"] += a.numel(); out["flips"] += nd; out["flips_outside_kink_set"] += outside; out["kink_set"] += int(near.sum())
        out["worst_forward_rel"] = max(out["worst_forward_rel"], fwd)
        if nd:
            out["worst_flip_pre"] = max(out["worst_flip_pre"], float(torch.maximum(pre[0][diff].abs(), pre[1][diff].abs()).max()))
        out["per_tensor"][k] = {"elements": a.numel(), "forward_rel": fwd, "flips": nd, "kink_set": int(near.sum())}
    return out


# ------------------------------------------------------------------------------------------------------------- config 5's calls against fp64 twins, config 4 against draws
def c5_call_vs_f64(dev, tag):
    """One call of BASELINE config 5's stream (fp32 storage; r4_cases.c5_call_case's call) against the reference's FP64 run of it (loop_c5_calls_f64.npz, round 5), with the
    reference's own fp32 evaluations of the same call beside it (loop_ref_draws.npz: `draw_*` lists, one entry per evaluation)."""
    from maxstyle_amd import synthetic as syn
    from test_solver_gpu import injector
    import r3_cases as R3
    import r4_cases as R4
    g32 = np.load(os.path.join(GOLDEN, "loop_c5_calls.npz"))
    g = np.load(os.path.join(GOLDEN, "loop_c5_calls_f64.npz"))
    dr = np.load(os.path.join(GOLDEN, "loop_ref_draws.npz"))
    c = R4.C5_CALLS[tag]
    spec = syn.NetSpec(*c["spec"])
    B, layers, K = 16, [3, 4, 5], int(g32[f"{tag}.K"])
    S = R3.trained_solver(dev, "trained_fcn16_256.npz") if tag == "acdc" else R4.trained_solver64(dev)
    img, lab = syn.synthetic_batch(B, c["size"], spec.image_ch, spec.num_classes, seed=int(g32[f"{tag}.seed"]))
    img_d, lab_d = img.to(dev), lab.to(dev)
    styles = {}
    for i, ap in zip(layers, [bool(v) for v in g32[f"{tag}.applied"]]):
        st = syn.random_style_state(B, spec.channel_num[i], 7 + i)
        st.perm = torch.from_numpy(g32[f"{tag}.{i}.perm"]).clone()
        st.applied = ap
        styles[i] = st
    S.style_init_hook = injector(styles, dev)
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    out = S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=0.5, n_iter=K, lr=0.1, reference_image=img_d, reference_segmentation=lab_d,
                                     fix_seed=int(g32[f"{tag}.fix_seed"]))
    losses = S.last_losses.cpu().numpy().astype(np.float64)
    scale = float(g[f"{tag}.image_scale"])
    o = out.cpu().double()
    d = o[:, :, ::4, ::4] - torch.from_numpy(g[f"{tag}.image.strided"]).double()
    pred = segment(S, out).argmax(1).cpu()
    ds = dice(pred, lab, spec.num_classes)
    return {"strided_max": float(d.abs().max()) / scale, "strided_rms": float(d.pow(2).mean().sqrt()) / scale,
            "mean_max": float(np.abs(o.mean(dim=(2, 3)).numpy() - g[f"{tag}.image.mean"]).max()) / scale,
            "rms_max": float(np.abs(o.pow(2).mean(dim=(2, 3)).sqrt().numpy() - g[f"{tag}.image.rms"]).max()) / scale,
            "losses_rel": (np.abs(losses - g[f"{tag}.losses"]) / np.abs(g[f"{tag}.losses"])).tolist(),
            "labels_equal": float((pred.numpy() == g[f"{tag}.final_pred"]).mean()),
            "dice_abs_diff": max(abs(a - float(b)) for a, b in zip(ds, g[f"{tag}.final_dice"])),
            "variants": [str(v) for v in dr["variants"]],
            "draw_strided_max": dr[f"{tag}.strided_max"].tolist(), "draw_strided_rms": dr[f"{tag}.strided_rms"].tolist(), "draw_mean_max": dr[f"{tag}.mean_max"].tolist(),
            "draw_rms_max": dr[f"{tag}.rms_max"].tolist(), "draw_losses_rel": dr[f"{tag}.losses_rel"].tolist(), "draw_labels_equal": dr[f"{tag}.labels_equal"].tolist()}


def c4_draws_all():
    """Every key of loop_ref_draws.npz as lists (`c2.*`, `c4.*`, `acdc.*`, `prostate.*`)."""
    dr = np.load(os.path.join(GOLDEN, "loop_ref_draws.npz"))
    return {k: dr[k].tolist() for k in dr.files if dr[k].dtype.kind == "f"}


def c4_draws():
    """The reference's own fp32 evaluations of the config-4 call against its fp64 run (loop_ref_draws.npz `c4.*`), for the bars of the config-4 test."""
    dr = np.load(os.path.join(GOLDEN, "loop_ref_draws.npz"))
    return {k[3:]: dr[k].tolist() for k in dr.files if k.startswith("c4.")} | {"variants": [str(v) for v in dr["variants"]]}


# ------------------------------------------------------------------------------------------------------------- the argument cases, teacher-forced
def arg_case_teacher_forced(dev, case):
    """Every step k of a drop-in argument case (tests/golden/loop_args.npz; "all6": loop_args_all6.npz) evaluated at the parameters and frozen batch std the reference's fp64
    run held before that step - its stored per-step gradients ARE fp64 evaluations at those points - so nothing free-running enters: per step the loss error and the gradient
    error of every learnable tensor (max norm over max|g|), beside the reference's own fp32 error at step 1 (the one step where its fp32 and fp64 runs share a point)."""
    from maxstyle_amd import synthetic as syn
    import r3_cases as R3
    g = np.load(os.path.join(GOLDEN, "loop_args_all6.npz" if case == "all6" else "loop_args.npz"))
    kw = dict(R3.ARG_CALLS.get(case, {}))
    B, layers = 4, ([0, 1, 2, 3, 4, 5] if case == "all6" else [3, 4, 5])
    spec = syn.NetSpec(4, 1, 4)
    S = R3.trained_solver(dev, "trained_fcn16.npz")
    img, lab = syn.synthetic_batch(B, 64, 1, 4, seed=777)
    img_d, lab_d = img.to(dev), lab.to(dev)
    drawn = case == "beta_drawn"
    names = [str(n) for n in g[f"{case}.param_names"]]
    per = len(kw.get("loss_weights", [1]))
    lw = float(sum(kw.get("loss_weights", [1])))
    K = len(g[f"{case}.f64.losses"]) // per
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    steps = []
    for k in range(1, K + 1):
        def hook(mods, k=k):
            for key, m in mods.items():
                i = int(key)
                if not drawn:
                    st = syn.random_style_state(B, spec.channel_num[i], 7 + i)
                    m.perm = st.perm.clone()
                    with torch.no_grad():
                        if isinstance(m.gamma_noise, torch.nn.Parameter):
                            m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev)
                        if isinstance(m.lmda, torch.nn.Parameter):
                            m.lmda.data = torch.from_numpy(g[f"{case}.initial.{i}.lmda"]).to(dev)
                if k > 1:
                    with torch.no_grad():
                        for nm in PN:
                            pk = f"{case}.f64.step{k - 1}.param.{i}.{nm}"
                            if pk in g.files and isinstance(getattr(m, nm), torch.nn.Parameter):
                                getattr(m, nm).data = torch.from_numpy(g[pk]).float().reshape(getattr(m, nm).shape).to(dev)
                    sk = f"{case}.f64.{i}.gamma_std"
                    if sk in g.files and bool(m.rand_p < m.p):
                        m.gamma_std = torch.from_numpy(g[sk]).float().reshape(1, -1, 1, 1).to(dev)
                        m.beta_std = torch.from_numpy(g[f"{case}.f64.{i}.beta_std"]).float().reshape(1, -1, 1, 1).to(dev)
        S.style_init_hook = hook
        call = dict(decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=1.5, lr=0.1, reference_image=img_d, reference_segmentation=lab_d)
        call.update(kw)
        call["n_iter"] = 1
        S.generate_max_style_image(z_i.detach(), **call)
        eng = next(iter(S._engines.values()))
        l64 = float(g[f"{case}.f64.losses"][::per][k - 1])
        ours, noise1 = {}, {}
        for n in names:
            gk = f"{case}.f64.step{k}.grad.{n}"
            if gk not in g.files:
                continue                                    # a Parameter without gradient (mix_learnable / noise_learnable False)
            i, nm = n.split(".")
            r64 = g[gk].astype(np.float64).reshape(-1)
            ours[n] = float(np.abs(eng.grad(int(i), nm).detach().cpu().numpy().astype(np.float64).reshape(-1) - r64).max() / max(np.abs(r64).max(), 1e-30))
            r1_64, r1_32 = g[f"{case}.f64.step1.grad.{n}"].astype(np.float64).reshape(-1), g[f"{case}.f32.step1.grad.{n}"].astype(np.float64).reshape(-1)
            noise1[n] = float(np.abs(r1_32 - r1_64).max() / max(np.abs(r1_64).max(), 1e-30))
        steps.append({"k": k, "loss_rel": abs(float(S.last_losses[0]) / lw - l64) / abs(l64), "ours": ours, "noise_step1": noise1})
    return {"winograd": bool(eng.winograd), "steps": steps}
