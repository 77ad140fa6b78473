"""Round-3 parity cases on the GPU, as functions that MEASURE and return numbers: tests/test_round3_gpu.py asserts on them, tools/parity_report.py
prints them (that is how the constants in the tests were chosen - from measured ratios against the reference's own fp32-vs-fp64 noise).

Fixtures: tests/golden/make_golden_r3.py (loop_full_c2.npz, loop_args.npz, loop_*_tf64.npz) - all produced by running the reference."""
import os

import numpy as np
import torch

from parity_util import rel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PN = ("gamma_noise", "beta_noise", "lmda")


def load_trained(name):
    z = np.load(os.path.join(GOLDEN, name))
    W = {"image_encoder": {}, "segmentation_decoder": {}, "image_decoder": {}}
    for key in z.files:
        net, k = key.split("/", 1)
        a = z[key]
        W[net][k] = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
    return W


def trained_solver(dev, weights):
    import maxstyle_amd as M
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True)
    W = load_trained(weights)
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)
        mod.train()
    return S


def segment(S, image):
    with torch.no_grad():
        _, zs = S.encode_image(image, disable_track_bn_stats=True)
        return S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs, disable_track_bn_stats=True)


def dice(pred, lab, ncls):
    out = []
    for c in range(1, ncls):
        a, b = pred == c, lab == c
        d = int(a.sum()) + int(b.sum())
        out.append(0.0 if d == 0 else 2.0 * int((a & b).sum()) / d)
    return out


# ------------------------------------------------------------------------------------------------------------- the benchmarked size
def full_size_case(dev):
    """generate_max_style_image through the drop-in solver at BASELINE config 2 (16x1x256x256, layers [3,4,5], K=5) on the trained FCN_16, against the
    REFERENCE's own fp64 run of exactly this call; `noise_*` = the reference's own fp32 run against that fp64 run.  The engine reads MS_LOOP_WINOGRAD
    when it is built, so the caller sets the environment before calling this."""
    from maxstyle_amd import synthetic as syn
    g = np.load(os.path.join(GOLDEN, "loop_full_c2.npz"))
    B, layers, K = 16, [3, 4, 5], 5
    spec = syn.NetSpec(4, 1, 4)
    S = trained_solver(dev, "trained_fcn16_256.npz")
    img, lab = syn.synthetic_batch(B, 256, 1, 4, seed=1234)
    img_d, lab_d = img.to(dev), lab.to(dev)
    styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}

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
    out = S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=K, lr=0.1, reference_image=img_d, reference_segmentation=lab_d)
    eng = next(iter(S._engines.values()))
    losses = S.last_losses.cpu().numpy().astype(np.float64)
    ref = torch.from_numpy(g["f64.image"]).double()
    scale = float(g["image_scale"])
    d = out.cpu().double() - ref
    pred = segment(S, out).argmax(1).cpu()
    clean_pred = segment(S, img_d).argmax(1).cpu()
    dsty, dclean = dice(pred, lab, 4), dice(clean_pred, lab, 4)
    res = {
        "winograd": bool(eng.winograd),
        "z_i_rel": rel(zf[idx], g["f64.z_i.sample"]),
        "image_max": float(d.abs().max()) / scale, "noise_image_max": float(g["ref_noise.image_max"]),
        "image_rms": float(d.pow(2).mean().sqrt()) / scale, "noise_image_rms": float(g["ref_noise.image_rms"]),
        "losses_rel": (np.abs(losses - g["f64.losses"]) / np.abs(g["f64.losses"])).tolist(), "noise_losses_rel": g["ref_noise.losses_rel"].tolist(),
        "losses": losses.tolist(),
        "labels_equal_f64": float((pred.numpy() == g["f64.final_pred"]).mean()), "noise_labels_equal": float(g["ref_noise.labels_equal"]),
        "clean_labels_equal": float((clean_pred.numpy() == g["f64.clean_pred"]).mean()),
        "dice": dsty, "dice_ref_f64": g["f64.final_dice"].tolist(), "dice_ref_f32": g["f32.final_dice"].tolist(),
        "dice_clean": dclean, "dice_clean_ref": g["f64.clean_dice"].tolist(),
        "dice_abs_diff": max(abs(a - b) for a, b in zip(dsty, g["f64.final_dice"])),
        "params_rel": {f"{i}.{nm}": rel(getattr(S.last_style_modules[str(i)], nm), g[f"f64.step{K}.param.{i}.{nm}"]) for i in layers for nm in PN},
        "noise_params_rel": {f"{i}.{nm}": rel(g[f"f32.step{K}.param.{i}.{nm}"], g[f"f64.step{K}.param.{i}.{nm}"]) for i in layers for nm in PN},
    }
    return res


# ------------------------------------------------------------------------------------------------------------- the arguments of the drop-in signature
# (the same table as tests/golden/make_golden_r3.py::ARG_CASES, restated: the generator cannot be imported on the GPU box - it imports the reference)
ARG_CALLS = {
    "nomix": dict(mix_style=False),
    "nonoise": dict(no_noise=True, noise_learnable=False),
    "mixfixed": dict(mix_learnable=False),
    "noisefixed": dict(noise_learnable=False),
    "lw05": dict(loss_weights=[0.5]),
    "twoterms": dict(loss_types=["seg", "seg"], loss_weights=[0.25, 0.5]),
    "k0": dict(n_iter=0),
    "lr003": dict(lr=0.03, n_iter=2),
    "beta_drawn": dict(always_use_beta=True, noise_learnable=False, fix_seed=11, p=0.8),
    "beta_injected": dict(always_use_beta=True),
}


def arg_case(dev, case, S=None):
    from maxstyle_amd import synthetic as syn
    # "all6": every decoder layer, on the trained network (its own fixture, tests/golden/make_golden_r3.py all6)
    g = np.load(os.path.join(GOLDEN, "loop_args_all6.npz" if case == "all6" else "loop_args.npz"))
    kw = dict(ARG_CALLS.get(case, {}))
    B, layers = 4, ([0, 1, 2, 3, 4, 5] if case == "all6" else [3, 4, 5])
    spec = syn.NetSpec(4, 1, 4)
    if S is None:
        S = trained_solver(dev, "trained_fcn16.npz")
    img, lab = syn.synthetic_batch(B, 64, 1, 4, seed=777)
    img_d, lab_d = img.to(dev), lab.to(dev)
    drawn = case == "beta_drawn"
    seen = {}

    def hook(mods):
        for k, m in mods.items():
            i = int(k)
            if not drawn:
                st = syn.random_style_state(B, spec.channel_num[i], 7 + i)
                m.perm = st.perm.clone()
                with torch.no_grad():
                    if isinstance(m.gamma_noise, torch.nn.Parameter):
                        m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev)
                    if isinstance(m.lmda, torch.nn.Parameter):
                        m.lmda.data = torch.from_numpy(g[f"{case}.initial.{i}.lmda"]).to(dev)
            seen[i] = {"perm": m.perm.clone(), "rand_p": m.rand_p.clone(), "applied": bool(m.rand_p < m.p),
                       **{nm: getattr(m, nm).detach().clone() for nm in PN}}
    S.style_init_hook = hook
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    call = dict(decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=1.5, n_iter=3, lr=0.1, reference_image=img_d, reference_segmentation=lab_d)
    call.update(kw)
    out = S.generate_max_style_image(z_i.detach(), **call)
    mods = S.last_style_modules
    lw = float(sum(kw.get("loss_weights", [1])))
    per = len(kw.get("loss_weights", [1]))
    ref64, ref32 = g[f"{case}.f64.losses"][::per], g[f"{case}.f32.losses"][::per]
    got = (S.last_losses.cpu().numpy().astype(np.float64) / lw) if S.last_losses is not None else np.zeros(0)
    res = {"z_i_rel": rel(z_i, g["z_i"]),
           "param_names": [f"{i}.{n}" for i in layers for n, _ in mods[str(i)].named_parameters()], "param_names_ref": [str(n) for n in g[f"{case}.param_names"]],
           "n_losses": (len(got), len(ref64)),
           "losses_rel": (np.abs(got - ref64) / np.abs(ref64)).tolist() if len(ref64) == len(got) else None,
           "noise_losses_rel": (np.abs(ref32 - ref64) / np.abs(ref64)).tolist(),
           "image_rel": rel(out, g[f"{case}.f64.image"]), "noise_image_rel": rel(g[f"{case}.f32.image"], g[f"{case}.f64.image"])}
    pred = segment(S, out).argmax(1).cpu()
    res["labels_equal"] = float((pred.numpy() == g[f"{case}.final_pred"]).mean())
    res["dice_abs_diff"] = max(abs(a - float(b)) for a, b in zip(dice(pred, lab, 4), g[f"{case}.final_dice"]))
    res["state_equal"] = all(np.array_equal(seen[i]["perm"].numpy(), g[f"{case}.f32.{i}.perm"]) and seen[i]["applied"] == bool(g[f"{case}.f32.{i}.applied"])
                             for i in layers)
    if drawn:
        res["rand_p_equal"] = all(np.array_equal(seen[i]["rand_p"].numpy(), g[f"{case}.f32.{i}.rand_p"]) for i in layers)
        res["drawn_lmda_equal"] = all(np.array_equal(seen[i]["lmda"].cpu().numpy(), g[f"{case}.initial.{i}.lmda"]) for i in layers if seen[i]["applied"])
    params, noise, fixed_ok = {}, {}, True
    learn = set(n for n in res["param_names"] if getattr(mods[n.split(".")[0]], n.split(".")[1]).requires_grad)
    for i in layers:
        if not seen[i]["applied"]:
            continue
        for nm in PN:
            if nm != "lmda" and kw.get("no_noise"):
                continue                                     # N(0,1) tensors the forward never reads
            cur = getattr(mods[str(i)], nm).detach()
            r64, r32 = g[f"{case}.f64.final.{i}.{nm}"], g[f"{case}.f32.final.{i}.{nm}"]
            if f"{i}.{nm}" in learn and call["n_iter"] > 0:
                params[f"{i}.{nm}"] = rel(cur, r64); noise[f"{i}.{nm}"] = rel(r32, r64)
            else:
                fixed_ok = fixed_ok and bool(torch.equal(cur.cpu().reshape(-1), seen[i][nm].cpu().reshape(-1))) and \
                    (float(np.abs(r64).max()) == 0.0 or rel(cur, r64) < 1e-6)
    res.update(params_rel=params, noise_params_rel=noise, fixed_params_unchanged=fixed_ok)
    return res


# ------------------------------------------------------------------------------------------------------------- teacher-forced steps with fp64 twins
def teacher_forced_noise_table(g, tf, layers, K):
    """Per step: the reference's own fp32 gradient error against its fp64 value at the same point, per tensor."""
    return {s: {f"{i}.{nm}": rel(g[f"step{s}.grad.{i}.{nm}"], tf[f"step{s}.grad.{i}.{nm}"]) for i in layers for nm in PN} for s in range(1, K + 1)}
