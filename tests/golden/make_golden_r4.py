"""Round-4 golden vectors, produced by running the REFERENCE here (needs /root/reference; older fixtures are left untouched).

    python tests/golden/make_golden_r4.py [train64] [full4] [c5]      (full4 / c5 need train64's output)

train64 -> trained_fcn64_320.npz : the three FCN_64 sub-nets (Prostate-shaped: 3 image channels, 2 classes - BASELINE config 4) trained by the reference's
           OWN training step (standard_training -> backward -> optimize_all_params, train_adv...py:163-199,532-535) on the synthetic stream: 64x64 from
           the procedural initialisation, then fine-tuned at the BENCHMARKED 320x320 (the synthetic anatomy scales with the resolution).  24.5 M
           parameters: conv weights are stored as int8 with one fp32 scale per output channel (w = q * scale; 25 MB instead of 49 MB in fp16), everything
           else fp16; BOTH sides load exactly the de-quantised values, so the fixture IS the network (the clean Dice printed below is measured on the
           stored values).
full4   -> loop_full_c4.npz : the reference's generate_max_style_image (advanced_triplet...py:458-571) at BASELINE config 4's size - 16x3x320x320, layers
           [3,4,5], K = 10, Adam lr 0.1 - on those weights, in fp32 and in fp64: losses, style parameters after every step, frozen gamma_std / beta_std,
           labels of the segmentation of the stylised image (uint8, all 16 samples), Dice, the reference's OWN fp32-vs-fp64 noise (image max / rms,
           losses, labels), and the fp64 image as fp32 - in full for samples 0, 5, 10, 15 (4.9 MB), as a [::4, ::4] strided sample plus per-(sample,
           channel) mean / rms in fp64 for all 16 (the whole image would be 19.7 MB).
c5      -> loop_c5_calls.npz : for BASELINE config 5's stream - one ACDC-shaped call (trained FCN_16 at 16x1x256x256, K = 5) and one Prostate-shaped call
           (trained FCN_64 at 16x3x320x320, K = 10) as the trainer issues them: p = 0.5, layer subset drawn under fix_seed on the CPU generator
           (maxstyle.py:54-73), noise injected - fp32 reference run: losses, image sample, labels, Dice.  The fp32 leg of the mixed stream and the
           bf16-storage oracle (oracle/maxstyle_oracle.py `store=`) are compared against these on the GPU box.
Fixtures are data only.  The reference is imported in place, never copied.
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from make_golden import inject  # noqa: E402
from make_golden_r3 import Spy, segment, sample_idx, NETS, PNAMES  # noqa: E402
from oracle import maxstyle_oracle as orc  # noqa: E402

SPEC4 = orc.NetSpec(1, 3, 2)
NET4 = dict(network_type="FCN_64_standard_no_STN", image_ch=3, num_classes=2)
CKPT = os.environ.get("MS_R4_CKPT", "/tmp/ms_r4_fcn64.pt")
NTHREADS = int(os.environ.get("MS_R4_THREADS", "8"))


# ----------------------------------------------------------------------------------------------------------------- storage
def quantise_state(models):
    """state_dicts -> npz dict: 4-D floating tensors (conv / transposed-conv weights) as int8 + per-output-channel fp32 scale, other floats fp16."""
    store = {}
    for n in NETS:
        for k, v in models[n].state_dict().items():
            a = v.detach().cpu().numpy()
            if not v.is_floating_point():
                store[f"{n}/{k}"] = a
            elif a.ndim == 4 and a.size >= 4096:
                s = np.abs(a).reshape(a.shape[0], -1).max(1).astype(np.float32) / 127.0
                s[s == 0] = 1.0
                q = np.clip(np.rint(a / s[:, None, None, None]), -127, 127).astype(np.int8)
                store[f"{n}/{k}"] = q
                store[f"{n}/{k}::scale"] = s
            else:
                # fp16 where it is exact enough and in range (running_var of a few layers exceeds 65504: those stay fp32)
                store[f"{n}/{k}"] = a.astype(np.float16) if float(np.abs(a).max(initial=0.0)) < 6.0e4 else a.astype(np.float32)
    return store


def dequantised(store, net, key):
    """The fp32 value of one stored tensor (the arithmetic both sides use: int8 * fp32 scale in fp32; fp16 widened)."""
    a = store[f"{net}/{key}"]
    if a.dtype == np.int8:
        return torch.from_numpy(a.astype(np.float32) * store[f"{net}/{key}::scale"][:, None, None, None])
    if a.dtype == np.float16:
        return torch.from_numpy(a.astype(np.float32))
    return torch.from_numpy(a)


def trained_reference64(solver_mod, dtype, **solver_kw):
    store = np.load(os.path.join(HERE, "trained_fcn64_320.npz"))
    with contextlib.redirect_stdout(io.StringIO()):
        R = solver_mod.AdvancedTripletReconSegmentationModel(use_gpu=False, **NET4, **solver_kw)
    for n in NETS:
        R.model[n].load_state_dict({k: dequantised(store, n, k) for k in R.model[n].state_dict()}, strict=True)
        R.model[n].train()
        if dtype == torch.float64:
            R.model[n].double()
    return R


# ----------------------------------------------------------------------------------------------------------------- training
def _train_phase(S, iters, B, size, seed0, every):
    for it in range(iters):
        t0 = time.time()
        clean, lab = orc.synthetic_batch(B, size, 3, 2, seed=seed0 + it)
        g = torch.Generator().manual_seed(seed0 + 4000 + it)
        noisy = torch.clamp(clean + 0.05 * torch.randn(clean.shape, generator=g), 0.0, 1.0)
        with contextlib.redirect_stdout(io.StringIO()):
            S.reset_all_optimizers()
            seg, rec, gt, sh = S.standard_training(clean, lab, perturbed_image=noisy)
            loss = seg + rec + gt + sh
            loss.backward()
            S.optimize_all_params()
        if it % every == 0 or it == iters - 1:
            print(f"train {size}x{size} it {it}: seg {float(seg):.4f} rec {float(rec):.5f}  ({time.time() - t0:.1f} s/it)", flush=True)


def train_64(solver_mod, it64=300, it320=90, more=0):
    """more > 0: continue from the checkpoint of an earlier call (MS_R4_CKPT) for `more` iterations at 320x320 (fresh AdamW moments, seeds 27000+).
    Every call appends what it ran to tests/golden/trained_fcn64_320.provenance.json (the continuation arguments behind the committed file were
    not recorded in round 4 - the sidecar says what is known; the fixture IS the network, both sides of every test load it, so parity does not depend on regenerating it)."""
    torch.set_num_threads(NTHREADS)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        S = solver_mod.AdvancedTripletReconSegmentationModel(use_gpu=False, optimizer_type="AdamW", learning_rate=(5e-4 if more else 1e-3), **NET4)
    if more:
        ck = torch.load(CKPT)
        for name in NETS:
            S.model[name].load_state_dict(ck[name], strict=True)
        S.train()
        _train_phase(S, more, 2, 320, 27000 + int(os.environ.get("MS_R4_SEED_OFF", "0")), 10)
    else:
        W = orc.procedural_weights(SPEC4, seed=0)
        for name, mod in S.model.items():
            mod.load_state_dict(W[name], strict=True)
        S.train()
        _train_phase(S, it64, 8, 64, 21000, 50)
        torch.save({n: S.model[n].state_dict() for n in NETS}, CKPT + ".64")
        for opt in S.optimizers.values():
            for gr in opt.param_groups:
                gr["lr"] = 5e-4
        _train_phase(S, it320, 2, 320, 26000, 10)
    torch.save({n: S.model[n].state_dict() for n in NETS}, CKPT)
    import json
    prov_path = os.path.join(HERE, "trained_fcn64_320.provenance.json")
    prov = json.load(open(prov_path)) if (more and os.path.exists(prov_path)) else {"file": "trained_fcn64_320.npz", "phases": []}
    if more:
        prov["phases"].append({"command": f"make_golden_r4.py more={more}", "size": 320, "iterations": more, "batch": 2, "lr": 5e-4, "seed0": 27000 + int(os.environ.get("MS_R4_SEED_OFF", "0")),
                               "MS_R4_SEED_OFF": os.environ.get("MS_R4_SEED_OFF", "0"), "adamw_moments": "fresh"})
    else:
        prov["phases"] = [{"command": "make_golden_r4.py train64", "size": 64, "iterations": it64, "batch": 8, "lr": 1e-3, "seed0": 21000, "init": "orc.procedural_weights(SPEC4, seed=0)"},
                          {"command": "(same call)", "size": 320, "iterations": it320, "batch": 2, "lr": 5e-4, "seed0": 26000}]
    prov["threads"] = NTHREADS
    json.dump(prov, open(prov_path, "w"), indent=1)
    store = quantise_state(S.model)
    path = os.path.join(HERE, "trained_fcn64_320.npz")
    np.savez_compressed(path, **store)
    img, lab = orc.synthetic_batch(4, 320, 3, 2, seed=1234)
    pred_fp = segment(S, img).argmax(1)
    R = trained_reference64(solver_mod, torch.float32)
    pred_q = segment(R, img).argmax(1)
    print("trained_fcn64_320.npz", os.path.getsize(path), "clean Dice (first 4 samples of the bench batch): trained fp32", orc.dice_per_class(pred_fp, lab, 2),
          "stored (int8 conv weights)", orc.dice_per_class(pred_q, lab, 2), flush=True)


# ----------------------------------------------------------------------------------------------------------------- config 4 at size
FULL_SAMPLES = (0, 5, 10, 15)


def full_c4(solver_mod, K=10):
    torch.set_num_threads(NTHREADS)
    B, size, layers = 16, 320, [3, 4, 5]
    img, lab = orc.synthetic_batch(B, size, 3, 2, seed=1234)          # the bench workload's batch (bench.py --config c4, rank 0)
    res = {"layers": np.array(layers), "K": np.array(K), "full_samples": np.array(FULL_SAMPLES)}
    images = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        t0 = time.time()
        R = trained_reference64(solver_mod, dtype)
        x = img.to(dtype)
        states = {i: orc.random_style_state(B, SPEC4.channel_num[i], 7 + i, dtype) for i in layers}
        with torch.no_grad():
            z_i, _ = R.encode_image(x, disable_track_bn_stats=True)
        Cpu = solver_mod.CpuMaxStyle
        Cpu.created = []
        Cpu.post_init_hook = staticmethod(lambda layer, idx: inject(layer, states[layers[idx]].clone(), dtype))
        with Spy(solver_mod) as spy, contextlib.redirect_stdout(io.StringIO()):
            out = R.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=SPEC4.channel_num, p=1.5, n_iter=K, lr=0.1,
                                             reference_image=x, reference_segmentation=lab)
        images[tag] = out
        clean_pred = segment(R, x).argmax(1)
        sty_pred = segment(R, out).argmax(1)
        res[f"{tag}.losses"] = np.array(spy.losses, np.float64)
        res[f"{tag}.clean_dice"] = np.array(orc.dice_per_class(clean_pred, lab, 2))
        res[f"{tag}.final_dice"] = np.array(orc.dice_per_class(sty_pred, lab, 2))
        res[f"{tag}.final_pred"] = sty_pred.numpy().astype(np.uint8)
        res[f"{tag}.clean_pred"] = clean_pred.numpy().astype(np.uint8)
        names = [f"{i}.{n}" for i in layers for n in PNAMES]
        for s_, ps in enumerate(spy.params):
            for n, p_ in zip(names, ps):
                res[f"{tag}.step{s_ + 1}.param.{n}"] = p_.numpy().astype(np.float32 if tag == "f32" else np.float64)
        for i, layer in zip(layers, Cpu.created):
            res[f"{tag}.{i}.gamma_std"] = layer.gamma_std.numpy().astype(np.float64)
            res[f"{tag}.{i}.beta_std"] = layer.beta_std.numpy().astype(np.float64)
        zf = z_i.reshape(-1)
        res[f"{tag}.z_i.sample"] = zf[sample_idx(zf.numel())].numpy().astype(np.float64)
        res[f"{tag}.z_i.stats"] = np.array([float(zf.mean()), float(zf.std()), float(zf.abs().max())])
        print(tag, f"({time.time() - t0:.0f} s)", "losses", spy.losses, "clean dice", res[f"{tag}.clean_dice"], "stylised dice", res[f"{tag}.final_dice"], flush=True)
        del R, spy
    i64 = images["f64"]
    i32 = images["f32"].double()
    res["f64.image.full"] = i64[list(FULL_SAMPLES)].numpy().astype(np.float32)
    res["f64.image.strided"] = i64[:, :, ::4, ::4].numpy().astype(np.float32)
    res["f64.image.mean"] = i64.mean(dim=(2, 3)).numpy()
    res["f64.image.rms"] = i64.pow(2).mean(dim=(2, 3)).sqrt().numpy()
    d = (i32 - i64)
    scale = float(i64.abs().max())
    res["image_scale"] = np.array(scale)
    res["ref_noise.image_max"] = np.array(float(d.abs().max()) / scale)
    res["ref_noise.image_rms"] = np.array(float(d.pow(2).mean().sqrt()) / scale)
    res["ref_noise.image_max_per_sample"] = (d.abs().amax(dim=(1, 2, 3)) / scale).numpy()
    res["ref_noise.image_rms_per_sample"] = (d.pow(2).mean(dim=(1, 2, 3)).sqrt() / scale).numpy()
    res["ref_noise.losses_rel"] = np.abs(res["f32.losses"] - res["f64.losses"]) / np.abs(res["f64.losses"])
    res["ref_noise.labels_equal"] = np.array(float((res["f32.final_pred"] == res["f64.final_pred"]).mean()))
    res["f32.image.strided"] = images["f32"][:, :, ::4, ::4].numpy()
    print("reference fp32-vs-fp64 noise at C4's size: image max", float(res["ref_noise.image_max"]), "rms", float(res["ref_noise.image_rms"]),
          "losses", res["ref_noise.losses_rel"], "labels equal", float(res["ref_noise.labels_equal"]), flush=True)
    path = os.path.join(HERE, "loop_full_c4.npz")
    np.savez_compressed(path, **res)
    print("loop_full_c4.npz", os.path.getsize(path), flush=True)


# ----------------------------------------------------------------------------------------------------------------- config 5 calls
def c5_calls(solver_mod):
    """One call per shape of the mixed stream, issued the way the trainer does (p = 0.5, fix_seed -> the layer subset and perm come from the CPU generator,
    maxstyle.py:54-73); the noise / lmda of the applied layers are injected (they come from the DEVICE generator in a GPU run)."""
    from make_golden_r3 import trained_reference
    from make_golden_r2 import inject_forced
    torch.set_num_threads(NTHREADS)
    res = {}
    for tag, mk, spec, size, K, seed, fix in (("acdc", lambda: trained_reference(solver_mod, torch.float32, "trained_fcn16_256.npz"), orc.NetSpec(4, 1, 4), 256, 5, 31001, 6),
                                              ("prostate", lambda: trained_reference64(solver_mod, torch.float32), SPEC4, 320, 10, 31002, 9)):      # fix_seed 6 -> layers {4, 5} applied, 9 -> {3, 5} (the reference's own draws under p = 0.5)
        B, layers = 16, [3, 4, 5]
        img, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=seed)
        R = mk()
        with torch.no_grad():
            z_i, _ = R.encode_image(img, disable_track_bn_stats=True)
        states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float32) for i in layers}
        Cpu = solver_mod.CpuMaxStyle
        Cpu.created = []

        def hook(layer, idx, states=states, layers=layers):
            # keep the reference's own draw of perm / rand_p (= which layers are applied); inject noise and lmda of the applied ones
            if "gamma_noise" in layer._parameters:
                st = states[layers[idx]]
                with torch.no_grad():
                    layer.gamma_noise.data = st.gamma_noise.clone(); layer.beta_noise.data = st.beta_noise.clone(); layer.lmda.data = st.lmda.clone()
        Cpu.post_init_hook = staticmethod(hook)
        with Spy(solver_mod) as spy, contextlib.redirect_stdout(io.StringIO()):
            out = R.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=0.5, n_iter=K, lr=0.1,
                                             reference_image=img, reference_segmentation=lab, fix_seed=fix)
        pred = segment(R, out).argmax(1)
        res[f"{tag}.seed"] = np.array(seed); res[f"{tag}.fix_seed"] = np.array(fix); res[f"{tag}.K"] = np.array(K)
        res[f"{tag}.losses"] = np.array(spy.losses, np.float64)
        res[f"{tag}.applied"] = np.array([bool(l.rand_p < l.p) for l in Cpu.created])
        for i, l in zip(layers, Cpu.created):
            res[f"{tag}.{i}.perm"] = l.perm.numpy(); res[f"{tag}.{i}.rand_p"] = l.rand_p.numpy()
        res[f"{tag}.image.strided"] = out[:, :, ::4, ::4].numpy()
        res[f"{tag}.image.mean"] = out.double().mean(dim=(2, 3)).numpy()
        res[f"{tag}.image.rms"] = out.double().pow(2).mean(dim=(2, 3)).sqrt().numpy()
        res[f"{tag}.image_scale"] = np.array(float(out.abs().max()))
        res[f"{tag}.final_pred"] = pred.numpy().astype(np.uint8)
        res[f"{tag}.final_dice"] = np.array(orc.dice_per_class(pred, lab, spec.num_classes))
        # the ORACLE (not the reference) with bf16 storage emulation on the same call: how far bf16 activation storage moves this loop at this size.
        # Labelled `oracle_bf16.*`; the GPU's bf16-storage run is compared with the reference run above inside a multiple of this distance.
        Wt = {n: {k: v.detach().clone() for k, v in R.model[n].state_dict().items()} for n in NETS}
        sty = {}
        for i, l in zip(layers, Cpu.created):
            st = states[i].clone()
            st.perm = l.perm.clone(); st.applied = bool(l.rand_p < l.p)
            sty[i] = st
        t0 = time.time()
        with orc.stored_as(orc.bf16_store):
            with torch.no_grad():
                zb, _ = orc.encoder_forward(Wt["image_encoder"], orc.bf16_store(img))
            tr = orc.InnerLoopTrace()
            ob = orc.generate_max_style_image(Wt, zb, sty, layers, lab, n_iter=K, lr=0.1, trace=tr)
        with torch.no_grad():
            zs = orc.encoder_forward(Wt["image_encoder"], ob)[1]
            pb = orc.decoder_forward(Wt["segmentation_decoder"], zs, "NN").argmax(1)
        scale = float(out.abs().max())
        res[f"{tag}.oracle_bf16.losses"] = np.array(tr.losses, np.float64)
        res[f"{tag}.oracle_bf16.image.strided"] = ob[:, :, ::4, ::4].numpy()
        res[f"{tag}.oracle_bf16.image_max"] = np.array(float((ob - out).abs().max()) / scale)
        res[f"{tag}.oracle_bf16.image_rms"] = np.array(float((ob - out).pow(2).mean().sqrt()) / scale)
        res[f"{tag}.oracle_bf16.labels_equal"] = np.array(float((pb == pred).float().mean()))
        res[f"{tag}.oracle_bf16.final_dice"] = np.array(orc.dice_per_class(pb, lab, spec.num_classes))
        print(tag, f"oracle with bf16 storage ({time.time() - t0:.0f} s): losses", tr.losses, "image max / rms vs the reference's fp32 run",
              float(res[f"{tag}.oracle_bf16.image_max"]), float(res[f"{tag}.oracle_bf16.image_rms"]), "labels equal", float(res[f"{tag}.oracle_bf16.labels_equal"]),
              "dice", res[f"{tag}.oracle_bf16.final_dice"], flush=True)
        print(tag, "applied", res[f"{tag}.applied"], "losses", spy.losses, "dice", res[f"{tag}.final_dice"], flush=True)
    path = os.path.join(HERE, "loop_c5_calls.npz")
    np.savez_compressed(path, **res)
    print("loop_c5_calls.npz", os.path.getsize(path), flush=True)


def main():
    what = sys.argv[1:] or ["train64", "full4", "c5"]
    solver_mod = ref_harness.load_solver_module()
    if "train64" in what:
        train_64(solver_mod)
    for w in what:
        if w.startswith("more="):
            train_64(solver_mod, more=int(w[5:]))
    if "full4" in what:
        full_c4(solver_mod)
    if "c5" in what:
        c5_calls(solver_mod)


if __name__ == "__main__":
    main()
