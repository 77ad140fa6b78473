"""Golden vectors for one OUTER training iteration (SURVEY 8(f)1,3), produced by running the REFERENCE here.

    python tests/golden/make_golden_outer.py        # needs /root/reference; writes tests/golden/outer_*.npz

The reference trainer's loop body (train_adv_supervised_segmentation_triplet.py:163-199, 251-287, 532-535) is driven through the
reference solver's own methods (standard_training, generate_max_style_image, hard_example_traininng, reset_all_optimizers,
optimize_all_params) with injected MaxStyle state, procedural weights and a stored input-noise tensor.  Full gradients / updated
weights are too large to commit: each tensor is stored as (L2 norm, sum, 64 strided samples); small tensors (<= 600 values) whole.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from make_golden import inject  # noqa: E402
from oracle import maxstyle_oracle as orc  # noqa: E402
from oracle import outer_oracle as outer  # noqa: E402

NETS = outer.NETS


def summarise(res, key, t):
    t = t.detach().double().reshape(-1).clone()
    res[key + ".norm"] = np.array(float(t.norm()))
    res[key + ".sum"] = np.array(float(t.sum()))
    if t.numel() <= 600:
        res[key + ".full"] = t.numpy()
    else:
        idx = torch.linspace(0, t.numel() - 1, 64).long()
        res[key + ".sample"] = t[idx].numpy()


def sample_idx(n):
    return torch.linspace(0, n - 1, 64).long()


def outer_case(solver_mod, spec, B, size, layers, K, dtype, optimizer="AdamW", n_outer=1):
    net = "FCN_16_standard_no_STN" if spec.reduce == 4 else "FCN_64_standard_no_STN"
    with contextlib.redirect_stdout(io.StringIO()):
        S = solver_mod.AdvancedTripletReconSegmentationModel(network_type=net, image_ch=spec.image_ch, num_classes=spec.num_classes,
                                                             use_gpu=False, optimizer_type=optimizer, learning_rate=1e-4)
    W = orc.procedural_weights(spec, seed=0)
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)
        if dtype == torch.float64:
            mod.double()
    clean, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
    clean = clean.to(dtype)
    chn = spec.channel_num
    Cpu = solver_mod.CpuMaxStyle
    res = {}
    # the oracle restatement runs beside the reference on its own copy of the weights
    OW = orc.procedural_weights(spec, seed=0, dtype=dtype)
    ostate = outer.new_optimizer_state(OW)
    for it in range(n_outer):
        g = torch.Generator().manual_seed(100 + it)
        noise = (0.05 * torch.randn(clean.shape, generator=g)).to(dtype)
        states = {i: orc.random_style_state(B, chn[i], 7 + i + 10 * it, dtype) for i in layers}
        ostyles = {i: s.clone() for i, s in states.items()}
        order = list(layers)
        Cpu.created = []
        Cpu.post_init_hook = staticmethod(lambda layer, idx: inject(layer, states[order[idx]].clone(), dtype))
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                S.train()
                S.reset_all_optimizers()
                image_l = torch.clamp(clean + noise, clean.min(), clean.max())
                seg0, rec0, gt0, sh0, recon0, p0, _ = S.standard_training(clean, lab, perturbed_image=image_l, return_output=True)
                z_i = S.z_i
                standard_loss = seg0 + rec0 + sh0 + gt0
                S.reset_all_optimizers()
                sty = S.generate_max_style_image(image_code=z_i, channel_num=chn, p=1.5, decoder_layers_indexes=list(layers), n_iter=K,
                                                 mix_style=True, lr=0.1, no_noise=False, reference_image=clean, reference_segmentation=lab,
                                                 noise_learnable=True, mix_learnable=True, loss_types=["seg"], loss_weights=[1])
                sty = sty.detach().clone()
                seg1, rec1, sh1, sh2 = S.hard_example_traininng(perturbed_image=sty, perturbed_seg=None, clean_image_l=clean, label_l=lab,
                                                                standard_input_image=image_l.detach().clone(), standard_recon_image=recon0)
                loss = standard_loss + (rec1 + seg1 + sh1 + sh2)
                S.reset_all_optimizers()
                loss.backward()
                grads = {f"{n}/{k}": p.grad.detach().clone() for n in NETS for k, p in S.model[n].named_parameters() if p.grad is not None}
                S.optimize_all_params()
        finally:
            Cpu.post_init_hook = None
        t = f"it{it}."
        res[t + "noise"] = noise.numpy()
        res[t + "losses"] = np.array([float(seg0), float(rec0), float(seg1), float(rec1)], np.float64)
        res[t + "stylised"] = sty.numpy() if size <= 32 else sty[:, :, ::4, ::4].numpy()
        summarise(res, t + "stylised.all", sty)
        summarise(res, t + "z_i", z_i)
        for k, gten in grads.items():
            summarise(res, t + "grad." + k, gten)
        for n in NETS:
            for k, v in S.model[n].state_dict().items():
                summarise(res, t + "after." + n + "/" + k, v)
        # ---- oracle beside it (same inputs) ----
        o = outer.train_iteration(OW, ostate, clean, lab, noise, ostyles, layers, n_iter=K, lr_inner=0.1, lr_outer=1e-4, optimizer=optimizer)
        dl = max(abs(a - b) for a, b in zip(res[t + "losses"], [o["seg_loss"], o["recon_loss"], o["hard_seg_loss"], o["hard_recon_loss"]]))
        live = [k for k in grads if not outer.is_null_grad_bias(*k.split("/", 1))]
        dg = max(float((o["grads"][k] - grads[k]).abs().max() / (grads[k].abs().max() + 1e-30)) for k in live)
        dw = max(float((OW[n][k].double() - v.double()).abs().max()) for n in NETS for k, v in S.model[n].state_dict().items()
                 if not outer.is_null_grad_bias(n, k))
        res[t + "null_grad_max"] = np.array(max(float(grads[k].abs().max()) for k in grads if k not in live))
        print(f"[{dtype}] it{it}: oracle vs reference: max|dloss|={dl:.3e} max rel grad diff={dg:.3e} max|dweight|={dw:.3e}", flush=True)
        res[t + "oracle_agreement"] = np.array([dl, dg, dw])
    return res


def main():
    solver_mod = ref_harness.load_solver_module()
    spec = orc.NetSpec(4, 1, 4)
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(HERE, "outer_small.npz"), **outer_case(solver_mod, spec, B=4, size=64, layers=[3, 4, 5], K=2, dtype=torch.float32, n_outer=2))
    np.savez_compressed(os.path.join(HERE, "outer_small_f64.npz"), **outer_case(solver_mod, spec, B=4, size=64, layers=[3, 4, 5], K=2, dtype=torch.float64, n_outer=2))
    np.savez_compressed(os.path.join(HERE, "outer_adam.npz"), **outer_case(solver_mod, spec, B=3, size=32, layers=[4], K=1, dtype=torch.float32, optimizer="Adam"))
    for f in ("outer_small.npz", "outer_small_f64.npz", "outer_adam.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
