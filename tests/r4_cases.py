"""Round-4 parity cases on the GPU (functions that MEASURE; tests/test_round4_gpu.py asserts on them, bench.py reports them): BASELINE config 4 at its
benchmarked size against the reference's own run, and config 5's two call shapes (p = 0.5, bf16 or fp32 activation storage) against the reference's fp32 run
with the bf16-storage ORACLE's distance from it as the calibration.  Fixtures: tests/golden/make_golden_r4.py."""
import os

import numpy as np
import torch

from parity_util import rel
from r3_cases import GOLDEN, PN, segment, dice, trained_solver

NETS = ("image_encoder", "segmentation_decoder", "image_decoder")


def load_trained64(name="trained_fcn64_320.npz"):
    """trained_fcn64_320.npz: conv weights int8 + one fp32 scale per output channel (w = q * scale, computed in fp32 exactly as the generator did), the rest fp16 / fp32."""
    z = np.load(os.path.join(GOLDEN, name))
    W = {n: {} for n in NETS}
    for key in z.files:
        if key.endswith("::scale"):
            continue
        net, k = key.split("/", 1)
        a = z[key]
        if a.dtype == np.int8:
            W[net][k] = torch.from_numpy(a.astype(np.float32) * z[key + "::scale"][:, None, None, None])
        else:
            W[net][k] = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
    return W


def trained_solver64(dev):
    import maxstyle_amd as M
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_64_standard_no_STN", image_ch=3, num_classes=2, use_gpu=True)
    W = load_trained64()
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)
        mod.train()
    return S


def _inject_all(styles, dev):
    def hook(mods):
        for k, m in mods.items():
            st = styles[int(k)]
            m.perm = st.perm.clone()
            with torch.no_grad():
                m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev); m.lmda.data = st.lmda.to(dev)
    return hook


def c4_full_case(dev):
    """generate_max_style_image through the drop-in solver at BASELINE config 4 (FCN_64, 16x3x320x320, layers [3,4,5], K = 10, Adam lr 0.1) on the trained FCN_64,
    against the REFERENCE's own fp64 run of exactly this call (loop_full_c4.npz); noise_* = the reference's own fp32 run against that fp64 run."""
    from maxstyle_amd import synthetic as syn
    g = np.load(os.path.join(GOLDEN, "loop_full_c4.npz"))
    B, layers, K = 16, [3, 4, 5], int(g["K"])
    spec = syn.NetSpec(1, 3, 2)
    S = trained_solver64(dev)
    img, lab = syn.synthetic_batch(B, 320, 3, 2, seed=1234)
    img_d, lab_d = img.to(dev), lab.to(dev)
    styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = _inject_all(styles, dev)
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    zf = z_i.detach().reshape(-1).cpu()
    idx = torch.linspace(0, zf.numel() - 1, 4096).long()
    out = S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=K, lr=0.1, reference_image=img_d, reference_segmentation=lab_d)
    eng = next(iter(S._engines.values()))
    losses = S.last_losses.cpu().numpy().astype(np.float64)
    scale = float(g["image_scale"])
    o = out.cpu().double()
    full = [int(i) for i in g["full_samples"]]
    d_full = o[full] - torch.from_numpy(g["f64.image.full"]).double()
    d_str = o[:, :, ::4, ::4] - torch.from_numpy(g["f64.image.strided"]).double()
    pred = segment(S, out).argmax(1).cpu()
    clean_pred = segment(S, img_d).argmax(1).cpu()
    dsty, dclean = dice(pred, lab, 2), dice(clean_pred, lab, 2)
    # the reference's own noise on the per-(sample, channel) moments: its fp32 run's strided sample against its fp64 run's (the same planes, every 4th pixel)
    r32, r64 = g["f32.image.strided"].astype(np.float64), g["f64.image.strided"].astype(np.float64)
    noise_plane_mean = float(np.abs((r32 - r64).mean(axis=(2, 3))).max()) / scale
    noise_plane_rms = float(np.abs(np.sqrt((r32 ** 2).mean(axis=(2, 3))) - np.sqrt((r64 ** 2).mean(axis=(2, 3)))).max()) / scale
    res = {
        "winograd": bool(eng.winograd), "K": K,
        "z_i_rel": rel(zf[idx], g["f64.z_i.sample"]),
        "image_max": max(float(d_full.abs().max()), float(d_str.abs().max())) / scale, "noise_image_max": float(g["ref_noise.image_max"]),
        "image_rms_full": float(d_full.pow(2).mean().sqrt()) / scale, "image_rms_strided": float(d_str.pow(2).mean().sqrt()) / scale,
        "noise_image_rms": float(g["ref_noise.image_rms"]),
        "mean_rel": float(np.abs(o.mean(dim=(2, 3)).numpy() - g["f64.image.mean"]).max()) / scale,
        "rms_rel": float(np.abs(o.pow(2).mean(dim=(2, 3)).sqrt().numpy() - g["f64.image.rms"]).max()) / scale,
        "noise_plane_mean": noise_plane_mean, "noise_plane_rms": noise_plane_rms,
        "losses": losses.tolist(),
        "losses_rel": (np.abs(losses - g["f64.losses"]) / np.abs(g["f64.losses"])).tolist(), "noise_losses_rel": g["ref_noise.losses_rel"].tolist(),
        "labels_equal_f64": float((pred.numpy() == g["f64.final_pred"]).mean()), "noise_labels_equal": float(g["ref_noise.labels_equal"]),
        "clean_labels_equal": float((clean_pred.numpy() == g["f64.clean_pred"]).mean()),
        "dice": dsty, "dice_ref_f64": g["f64.final_dice"].tolist(), "dice_ref_f32": g["f32.final_dice"].tolist(),
        "dice_clean": dclean, "dice_clean_ref": g["f64.clean_dice"].tolist(),
        "dice_abs_diff": max(abs(a - b) for a, b in zip(dsty, g["f64.final_dice"])),
        "params_rel": {f"{i}.{nm}": rel(getattr(S.last_style_modules[str(i)], nm), g[f"f64.step{K}.param.{i}.{nm}"]) for i in layers for nm in PN},
        "noise_params_rel": {f"{i}.{nm}": rel(g[f"f32.step{K}.param.{i}.{nm}"], g[f"f64.step{K}.param.{i}.{nm}"]) for i in layers for nm in PN},
    }
    return res


C5_CALLS = {"acdc": dict(spec=(4, 1, 4), size=256), "prostate": dict(spec=(1, 3, 2), size=320)}


def c5_call_case(dev, tag, act_dtype=None):
    """One call of BASELINE config 5's mixed stream as the trainer issues it (p = 0.5: a strict subset of [3,4,5] applied - the reference's own draw under
    fix_seed, injected here together with its perm and the noise), on fp32 or bf16 activation storage, against the REFERENCE's fp32 run of the same call
    (loop_c5_calls.npz); oracle_bf16_* = how far the CPU oracle with bf16 storage emulation lands from that run (the calibration of the bf16 bars)."""
    from maxstyle_amd import synthetic as syn
    from test_solver_gpu import injector
    g = np.load(os.path.join(GOLDEN, "loop_c5_calls.npz"))
    c = C5_CALLS[tag]
    spec = syn.NetSpec(*c["spec"])
    B, layers, K = 16, [3, 4, 5], int(g[f"{tag}.K"])
    S = trained_solver(dev, "trained_fcn16_256.npz") if tag == "acdc" else trained_solver64(dev)
    S.loop_act_dtype = act_dtype
    img, lab = syn.synthetic_batch(B, c["size"], spec.image_ch, spec.num_classes, seed=int(g[f"{tag}.seed"]))
    img_d, lab_d = img.to(dev), lab.to(dev)
    applied = [bool(v) for v in g[f"{tag}.applied"]]
    styles = {}
    for i, ap in zip(layers, applied):
        st = syn.random_style_state(B, spec.channel_num[i], 7 + i)
        st.perm = torch.from_numpy(g[f"{tag}.{i}.perm"]).clone()
        st.applied = ap
        styles[i] = st
    S.style_init_hook = injector(styles, dev)
    z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
    out = S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=0.5, n_iter=K, lr=0.1, reference_image=img_d, reference_segmentation=lab_d,
                                     fix_seed=int(g[f"{tag}.fix_seed"]))
    eng = next(iter(S._engines.values()))
    losses = S.last_losses.cpu().numpy().astype(np.float64)
    scale = float(g[f"{tag}.image_scale"])
    o = out.cpu().double()
    d = o[:, :, ::4, ::4] - torch.from_numpy(g[f"{tag}.image.strided"]).double()
    d_or = torch.from_numpy(g[f"{tag}.oracle_bf16.image.strided"]).double() - torch.from_numpy(g[f"{tag}.image.strided"]).double()      # (same pixel subset as d)
    S.loop_act_dtype = None
    pred = segment(S, out).argmax(1).cpu()
    ds = dice(pred, lab, spec.num_classes)
    return {
        "applied": sorted(eng.layers), "applied_ref": [i for i, ap in zip(layers, applied) if ap], "storage": str(eng.act_dtype),
        "losses": losses.tolist(), "losses_ref": g[f"{tag}.losses"].tolist(),
        "losses_rel": (np.abs(losses - g[f"{tag}.losses"]) / np.abs(g[f"{tag}.losses"])).tolist(),
        "oracle_bf16_losses_rel": (np.abs(g[f"{tag}.oracle_bf16.losses"] - g[f"{tag}.losses"]) / np.abs(g[f"{tag}.losses"])).tolist(),
        "image_max": float(d.abs().max()) / scale, "image_rms": float(d.pow(2).mean().sqrt()) / scale,
        "mean_rel": float(np.abs(o.mean(dim=(2, 3)).numpy() - g[f"{tag}.image.mean"]).max()) / scale,
        "oracle_bf16_image_max": float(d_or.abs().max()) / scale, "oracle_bf16_image_rms": float(d_or.pow(2).mean().sqrt()) / scale,
        "labels_equal": float((pred.numpy() == g[f"{tag}.final_pred"]).mean()), "oracle_bf16_labels_equal": float(g[f"{tag}.oracle_bf16.labels_equal"]),
        "dice": ds, "dice_ref": g[f"{tag}.final_dice"].tolist(), "oracle_bf16_dice": g[f"{tag}.oracle_bf16.final_dice"].tolist(),
        "dice_abs_diff": max(abs(a - b) for a, b in zip(ds, g[f"{tag}.final_dice"])),
    }
