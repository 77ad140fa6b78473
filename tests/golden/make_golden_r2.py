"""Round-2 golden vectors, produced by running the REFERENCE here (needs /root/reference; the older fixtures are left untouched).

    python tests/golden/make_golden_r2.py [depth] [trained] [c4]

depth   -> loop_random_depth.npz : generate_max_style_image as the TRAINER calls it - p = 0.5 literal
           (train_adv_supervised_segmentation_triplet.py:263) - with injected rand_p so that strict subsets of the inserted layers
           [3,4,5] are applied: {3}, {4,5}, {} (the not-applied layers take MaxStyle's identity path, maxstyle.py:62-73,146-152).
trained -> trained_fcn16.npz     : the three FCN_16 sub-nets after a short run of the reference's OWN training step
           (standard_training -> backward -> optimize_all_params, train_adv...py:163-199,532-535) on the synthetic ACDC-shaped
           stream, rounded to fp16 for storage (both sides load exactly these values);
           loop_trained.npz      : the reference's K=5 MaxStyle loop on those weights (fp32 + fp64 twin): losses, final image,
           segmentation of the clean and of the stylised image and their Dice against the labels.  With trained networks the Dice
           is a meaningful number (clean ~0.9, stylised lower: the hard example), unlike the ~0.07 of random networks.
c4      -> loop_c4small.npz (+ _f64): the reference's loop on the Prostate-shaped FCN_64 / 3-channel / 2-class network (BASELINE config 4) at 4x3x64x64.
Fixtures are data only.  The reference is imported in place, never copied.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from make_golden import inject, build_reference_solver  # noqa: E402
from oracle import maxstyle_oracle as orc  # noqa: E402

NETS = ("image_encoder", "segmentation_decoder", "image_decoder")


def inject_forced(layer, st, dtype):
    """`inject` for layers built with the trainer's p = 0.5: the construction-time draw (rand_p < p or not, maxstyle.py:62-73) decides whether the
    layer owns Parameters at all, so an "applied" state can only be injected into a layer that drew "applied" - re-draw with p = 2 when it did not."""
    if st.applied and "gamma_noise" not in layer._parameters:
        p0, layer.p = layer.p, 2.0
        layer.init_parameters()
        layer.p = p0
    inject(layer, st, dtype)


def run_loop(solver_mod, S, spec, img, lab, layers, states, K, dtype, p, lr=0.1):
    """Reference generate_max_style_image with injected MaxStyle state; returns (image, losses, per-step parameters of the applied layers)."""
    with torch.no_grad():
        z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
    Cpu = solver_mod.CpuMaxStyle
    Cpu.created = []
    order = list(layers)
    Cpu.post_init_hook = staticmethod(lambda layer, idx: inject_forced(layer, states[order[idx]].clone(), dtype))
    losses, params_trace = [], []
    orig_step = torch.optim.Adam.step

    def step_spy(self, *a, **k):
        r = orig_step(self, *a, **k)
        params_trace.append([p_.detach().clone() for g in self.param_groups for p_ in g["params"]])
        return r
    torch.optim.Adam.step = step_spy
    orig_loss = solver_mod.basic_loss_fn

    def loss_spy(*a, **k):
        l = orig_loss(*a, **k)
        losses.append(-float(l))
        return l
    solver_mod.basic_loss_fn = loss_spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = S.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=p, n_iter=K,
                                             lr=lr, reference_image=img, reference_segmentation=lab)
    finally:
        torch.optim.Adam.step = orig_step
        solver_mod.basic_loss_fn = orig_loss
        Cpu.post_init_hook = None
    return z_i, out, losses, params_trace


def segment(S, image):
    with torch.no_grad():
        _, zs = S.encode_image(image, disable_track_bn_stats=True)
        return S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs, disable_track_bn_stats=True)


def random_depth(solver_mod):
    torch.set_num_threads(1)
    spec = orc.NetSpec(4, 1, 4)
    B, size, layers, K = 4, 64, [3, 4, 5], 2
    res = {}
    for tag, applied in (("only3", {3}), ("l45", {4, 5}), ("none", set())):
        S, W = build_reference_solver(solver_mod, spec, torch.float32)
        img, lab = orc.synthetic_batch(B, size, 1, 4, seed=1234)
        states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float32, applied=(i in applied)) for i in layers}
        z_i, out, losses, ptrace = run_loop(solver_mod, S, spec, img, lab, layers, states, K, torch.float32, p=0.5)
        res[f"{tag}.image"] = out.numpy()
        res[f"{tag}.losses"] = np.array(losses, np.float64)
        res[f"{tag}.applied"] = np.array(sorted(applied), np.int64)
        names = [f"{i}.{n}" for i in layers if i in applied for n in ("gamma_noise", "beta_noise", "lmda")]
        for s_, ps in enumerate(ptrace):
            assert len(ps) == len(names), (tag, len(ps), names)
            for n, p_ in zip(names, ps):
                res[f"{tag}.step{s_ + 1}.param.{n}"] = p_.numpy()
        pred = segment(S, out).argmax(1)
        res[f"{tag}.final_dice"] = np.array(orc.dice_per_class(pred, lab, 4))
        print(tag, "losses", losses, "dice", res[f"{tag}.final_dice"], flush=True)
    np.savez_compressed(os.path.join(HERE, "loop_random_depth.npz"), **res)


def train_reference(solver_mod, iters=400, B=8, size=64, lr=1e-3):
    """The reference's own training step on the synthetic stream (a new batch per iteration, seeds 5000+it)."""
    torch.set_num_threads(8)
    torch.manual_seed(0)
    spec = orc.NetSpec(4, 1, 4)
    with contextlib.redirect_stdout(io.StringIO()):
        S = solver_mod.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=False,
                                                             optimizer_type="AdamW", learning_rate=lr)
    W = orc.procedural_weights(spec, seed=0)
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)
    S.train()
    for it in range(iters):
        clean, lab = orc.synthetic_batch(B, size, 1, 4, seed=5000 + it)
        g = torch.Generator().manual_seed(9000 + it)
        noisy = torch.clamp(clean + 0.05 * torch.randn(clean.shape, generator=g), 0.0, 1.0)
        with contextlib.redirect_stdout(io.StringIO()):
            S.reset_all_optimizers()
            seg, rec, gt, sh = S.standard_training(clean, lab, perturbed_image=noisy)
            loss = seg + rec + gt + sh
            loss.backward()
            S.optimize_all_params()
        if it % 50 == 0 or it == iters - 1:
            print(f"train it {it}: seg {float(seg):.4f} rec {float(rec):.5f}", flush=True)
    return S


def trained(solver_mod):
    spec = orc.NetSpec(4, 1, 4)
    S = train_reference(solver_mod)
    # storage: floating-point tensors as fp16 (both sides then load exactly these values), integer buffers as they are
    store = {}
    for n in NETS:
        for k, v in S.model[n].state_dict().items():
            store[f"{n}/{k}"] = v.numpy().astype(np.float16) if v.is_floating_point() else v.numpy()
    np.savez_compressed(os.path.join(HERE, "trained_fcn16.npz"), **store)
    print("trained_fcn16.npz", os.path.getsize(os.path.join(HERE, "trained_fcn16.npz")), flush=True)
    B, size, layers, K = 4, 64, [3, 4, 5], 5
    img, lab = orc.synthetic_batch(B, size, 1, 4, seed=777)           # a batch the training stream never produced
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_num_threads(1)
        with contextlib.redirect_stdout(io.StringIO()):
            R = solver_mod.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=False)
        for n in NETS:
            sd = {k: torch.from_numpy(store[f"{n}/{k}"].astype(np.float32) if store[f"{n}/{k}"].dtype == np.float16 else store[f"{n}/{k}"])
                  for k in R.model[n].state_dict()}
            R.model[n].load_state_dict(sd, strict=True)
            R.model[n].train()
            if dtype == torch.float64:
                R.model[n].double()
        x = img.to(dtype)
        states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, dtype) for i in layers}
        z_i, out, losses, ptrace = run_loop(solver_mod, R, spec, x, lab, layers, states, K, dtype, p=1.5)
        clean_pred = segment(R, x).argmax(1)
        sty_logits = segment(R, out)
        sty_pred = sty_logits.argmax(1)
        res[f"{tag}.image"] = out.numpy()
        res[f"{tag}.losses"] = np.array(losses, np.float64)
        res[f"{tag}.z_i"] = z_i.numpy()
        res[f"{tag}.clean_dice"] = np.array(orc.dice_per_class(clean_pred, lab, 4))
        res[f"{tag}.final_dice"] = np.array(orc.dice_per_class(sty_pred, lab, 4))
        res[f"{tag}.final_pred"] = sty_pred.numpy().astype(np.uint8)
        names = [f"{i}.{n}" for i in layers for n in ("gamma_noise", "beta_noise", "lmda")]
        for n, p_ in zip(names, ptrace[-1]):
            res[f"{tag}.step{K}.param.{n}"] = p_.numpy()
        print(tag, "losses", losses, "clean dice", res[f"{tag}.clean_dice"], "stylised dice", res[f"{tag}.final_dice"], flush=True)
    res["fp32_vs_fp64_image_rel"] = np.array(float(np.abs(res["f32.image"] - res["f64.image"]).max() / np.abs(res["f64.image"]).max()))
    print("reference fp32 vs fp64 image noise:", float(res["fp32_vs_fp64_image_rel"]), flush=True)
    np.savez_compressed(os.path.join(HERE, "loop_trained.npz"), **res)
    print("loop_trained.npz", os.path.getsize(os.path.join(HERE, "loop_trained.npz")), flush=True)


def config4_small(solver_mod):
    """loop_c4small.npz (+ fp64 twin): the reference's own loop on the Prostate-shaped network of BASELINE config 4 - FCN_64 widths (512-channel code,
    decoder 256/128/64/64), 3 image channels, 2 classes (advanced_triplet...py:42,152-171; channel_num = [512,256,128,64,64,3], train_adv...py:255-258) -
    at 4x3x64x64, MaxStyle after blocks [3,4,5], K = 3.  Same keys as loop_c2small.npz (make_golden.loop_case)."""
    from make_golden import loop_case
    torch.set_num_threads(8)
    spec = orc.NetSpec(1, 3, 2)
    c = loop_case(solver_mod, spec, B=4, size=64, layers=[3, 4, 5], K=3, dtype=torch.float32)
    np.savez_compressed(os.path.join(HERE, "loop_c4small.npz"), **c)
    d = loop_case(solver_mod, spec, B=4, size=64, layers=[3, 4, 5], K=3, dtype=torch.float64)
    np.savez_compressed(os.path.join(HERE, "loop_c4small_f64.npz"), **{k: v for k, v in d.items() if k in ("image", "losses") or k.startswith("step")})
    print("c4small losses", c["losses"], "dice", c["final_dice"], flush=True)


def main():
    what = sys.argv[1:] or ["depth", "trained"]
    solver_mod = ref_harness.load_solver_module()
    if "c4" in what:
        config4_small(solver_mod)
    if "depth" in what:
        random_depth(solver_mod)
    if "trained" in what:
        trained(solver_mod)


if __name__ == "__main__":
    main()
