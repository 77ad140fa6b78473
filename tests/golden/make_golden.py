"""Generate golden vectors by IMPORTING the reference (runs only where /root/reference exists).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures are data only (inputs that cannot be re-derived + expected outputs). Inputs that are a
closed-form function of a seed (procedural weights, synthetic batch, injected style state) are
rebuilt by the tests through oracle/maxstyle_oracle.py helpers, which this script uses as well so
both sides see identical inputs.  The reference is never copied: it is imported in place.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from oracle import maxstyle_oracle as orc  # noqa: E402

torch.set_num_threads(1)  # bit-reproducible fixtures (results move 1e-4 with thread count)


def inject(ref_layer, st: orc.StyleState, dtype):
    """Overwrite a reference MaxStyle instance's random state with ours."""
    ref_layer.perm = st.perm.clone()
    ref_layer.rand_p = torch.tensor([0.0 if st.applied else 1.0])
    if st.applied:
        with torch.no_grad():
            ref_layer.gamma_noise.data = st.gamma_noise.detach().clone().to(dtype)
            ref_layer.beta_noise.data = st.beta_noise.detach().clone().to(dtype)
            ref_layer.lmda.data = st.lmda.detach().clone().to(dtype)
    else:
        # mimic the "not applied" branch of init_parameters (maxstyle.py:62-73)
        for n in ("gamma_noise", "beta_noise", "lmda"):
            if n in ref_layer._parameters:
                del ref_layer._parameters[n]
        B, C = ref_layer.batch_size, ref_layer.num_feature
        ref_layer.gamma_noise = torch.zeros(B, C, 1, 1, dtype=dtype)
        ref_layer.beta_noise = torch.zeros(B, C, 1, 1, dtype=dtype)
        ref_layer.lmda = torch.zeros(B, 1, 1, 1, dtype=dtype)


def known_answer_ramp(MaxStyle):
    """The reference's own __main__ smoke (maxstyle.py:193-241) with its RNG draws pinned."""
    torch.manual_seed(43)
    feats = (3 * torch.arange(32, dtype=torch.float32) + 5).view(4, 2, 2, 2)
    aug = MaxStyle(batch_size=4, num_feature=2, p=0.5, mix_style=True, mix_learnable=True, noise_learnable=True,
                   always_use_beta=False, no_noise=False, use_gpu=False)
    out = {"features": feats.numpy(), "perm": aug.perm.numpy(), "rand_p": aug.rand_p.numpy(),
           "gamma_noise0": aug.gamma_noise.detach().numpy().copy(), "beta_noise0": aug.beta_noise.detach().numpy().copy(),
           "lmda0": aug.lmda.detach().numpy().copy()}
    opt = torch.optim.Adam(list(aug.parameters()), lr=0.1)
    losses, outs = [], []
    for i in range(5):
        y = aug(feats)
        loss = torch.nn.MSELoss()(y, torch.ones_like(feats))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item()); outs.append(y.detach().numpy().copy())
    out.update(losses=np.array(losses, np.float64), outputs=np.stack(outs), gamma_std=aug.gamma_std.numpy(),
               beta_std=aug.beta_std.numpy(), gamma_noise5=aug.gamma_noise.detach().numpy().copy(),
               beta_noise5=aug.beta_noise.detach().numpy().copy(), lmda5=aug.lmda.detach().numpy().copy())
    return out


def layer_case(MaxStyle, B, C, H, W, seed, dtype, lmda_outside=False, mix_style=True):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, C, H, W, generator=g) * torch.rand(B, C, 1, 1, generator=g) * 2 + torch.randn(B, C, 1, 1, generator=g)).to(dtype)
    dy = torch.randn(B, C, H, W, generator=g).to(dtype)
    st = orc.random_style_state(B, C, seed + 1, dtype)
    if lmda_outside:
        st.lmda[0] = 1.3
        st.lmda[-1] = -0.2
    st.mix_style = mix_style
    layer = MaxStyle(B, C, p=1.5, mix_style=mix_style, use_gpu=False)  # p>1: always built 'applied'
    inject(layer, st, dtype)
    xr = x.clone().requires_grad_(True)
    y = layer(xr)
    y.backward(dy)
    res = {"x": x.numpy(), "dy": dy.numpy(), "perm": st.perm.numpy(), "lmda": st.lmda.numpy(),
           "gamma_noise": st.gamma_noise.numpy(), "beta_noise": st.beta_noise.numpy(),
           "y": y.detach().numpy(), "dx": xr.grad.numpy(), "gamma_std": layer.gamma_std.numpy(),
           "beta_std": layer.beta_std.numpy(), "d_gamma": layer.gamma_noise.grad.numpy(),
           "d_beta": layer.beta_noise.grad.numpy()}
    if mix_style:
        res["d_lmda"] = layer.lmda.grad.numpy()
    return res


def mixstyle_cases():
    """Reference MixStyle / DSU (src/advanced/mixstyle.py) with its RNG draws replayed and recorded."""
    ref_harness.install()
    from src.advanced.mixstyle import MixStyle as RefMix
    out = {}
    for tag, mix, lm in (("random", "random", None), ("cross", "crossdomain", None), ("extrap", "random", 1.7), ("dsu", "gaussian", None)):
        B, C, H, W = 6, 5, 12, 10
        g = torch.Generator().manual_seed(21)
        x = torch.randn(B, C, H, W, generator=g) * 0.7 + torch.randn(B, C, 1, 1, generator=g)
        dy = torch.randn(B, C, H, W, generator=g)
        layer = RefMix(p=1.0, alpha=0.1, mix=mix, lmda=lm)
        torch.manual_seed(5)
        xr = x.clone().requires_grad_(True)
        y = layer(xr)
        y.backward(dy)
        # replay the draws in the reference's order to record them
        torch.manual_seed(5)
        torch.rand(1)
        rec = {}
        if lm is None:
            rec["lmda"] = layer.beta.sample((B, 1, 1, 1)).numpy()
        if mix == "gaussian":
            rec["gaussian_mu"] = torch.randn(B, C, 1, 1).numpy(); rec["gaussian_std"] = torch.randn(B, C, 1, 1).numpy()
        else:
            rec["perm"] = layer.perm.numpy()
        out.update({f"{tag}.x": x.numpy(), f"{tag}.dy": dy.numpy(), f"{tag}.y": y.detach().numpy(), f"{tag}.dx": xr.grad.numpy()})
        out.update({f"{tag}.{k}": v for k, v in rec.items()})
    return out


def build_reference_solver(solver_mod, spec: orc.NetSpec, dtype):
    net = "FCN_16_standard_no_STN" if spec.reduce == 4 else "FCN_64_standard_no_STN"
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        S = solver_mod.AdvancedTripletReconSegmentationModel(network_type=net, image_ch=spec.image_ch,
                                                             num_classes=spec.num_classes, use_gpu=False)
    W = orc.procedural_weights(spec, seed=0)
    for name, mod in S.model.items():
        missing = mod.load_state_dict(W[name], strict=True)
        mod.train()
        if dtype == torch.float64:
            mod.double()
    return S, W


def loop_case(solver_mod, spec, B, size, layers, K, dtype, style_seed=7, lr=0.1, with_taps=False, eval_mode=False):
    """Runs the reference generate_max_style_image with injected state; returns expected outputs.
    eval_mode: the sub-networks are in .eval() (test-time use of the loop): every BatchNorm normalises with its running statistics."""
    S, W = build_reference_solver(solver_mod, spec, dtype)
    if eval_mode:
        for mod in S.model.values():
            mod.eval()
    img, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
    img = img.to(dtype)
    with torch.no_grad():
        z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
    chn = spec.channel_num
    states = {i: orc.random_style_state(B, chn[i], style_seed + i, dtype) for i in layers}
    Cpu = solver_mod.CpuMaxStyle
    Cpu.created = []
    order = list(layers)

    def hook(layer, idx):
        inject(layer, states[order[idx]].clone(), dtype)
    Cpu.post_init_hook = staticmethod(hook)

    # per-step capture through Adam hooks
    losses, params_trace, grads_trace = [], [], []
    orig_step = torch.optim.Adam.step

    def step_spy(self, *a, **k):
        grads_trace.append([None if p.grad is None else p.grad.detach().clone() for g in self.param_groups for p in g["params"]])
        r = orig_step(self, *a, **k)
        params_trace.append([p.detach().clone() for g in self.param_groups for p in g["params"]])
        return r
    torch.optim.Adam.step = step_spy
    orig_loss = solver_mod.basic_loss_fn

    def loss_spy(*a, **k):
        l = orig_loss(*a, **k)
        losses.append(-float(l))
        return l
    solver_mod.basic_loss_fn = loss_spy
    import io, contextlib
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = S.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=chn, p=1.5, n_iter=K,
                                             lr=lr, reference_image=img, reference_segmentation=lab)
    finally:
        torch.optim.Adam.step = orig_step
        solver_mod.basic_loss_fn = orig_loss
        Cpu.post_init_hook = None
    res = {"image": out.numpy(), "losses": np.array(losses, np.float64), "z_i": z_i.numpy()}
    names = []
    for i in layers:
        names += [f"{i}.gamma_noise", f"{i}.beta_noise", f"{i}.lmda"]
    for s, (ps, gs) in enumerate(zip(params_trace, grads_trace)):
        for n, p, g in zip(names, ps, gs):
            res[f"step{s+1}.param.{n}"] = p.numpy()
            if g is not None:
                res[f"step{s+1}.grad.{n}"] = g.numpy()
    for i, layer in zip(layers, Cpu.created):
        res[f"{i}.gamma_std"] = layer.gamma_std.numpy()
        res[f"{i}.beta_std"] = layer.beta_std.numpy()
    # segmentation of the final stylised image (Dice parity material): train-mode batch-stat forward
    with torch.no_grad():
        zi2, zs2 = S.encode_image(out, disable_track_bn_stats=True)
        logits = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs2, disable_track_bn_stats=True)
    pred = logits.argmax(1)
    res["final_logits_absmean"] = np.array(float(logits.abs().mean()))
    res["final_pred"] = pred.numpy().astype(np.uint8)
    res["final_dice"] = np.array(orc.dice_per_class(pred, lab, spec.num_classes))
    if with_taps:
        # per-block activations of the plain (no style) decoder / encoder / seg decoder in batch-stat mode
        with torch.no_grad():
            taps = {}
            dec = S.model["image_decoder"]
            from src.models.model_util import _disable_tracking_bn_stats as dtbs
            h = z_i
            for k in range(1, 5):
                blk = getattr(dec, f"up{k}")
                with dtbs(blk):
                    h = blk(h)
                taps[f"dec.up{k}"] = h
            recon0 = torch.sigmoid(dec.final_conv(h))
            taps["dec.recon"] = recon0
            enc = S.model["image_encoder"].general_encoder
            with dtbs(enc):
                x1 = torch.nn.functional.leaky_relu(enc.inc(recon0), 0.2)
                taps["enc.inc"] = x1
                h = x1
                for k in range(1, 5):
                    h = getattr(enc, f"down{k}")(h)
                    taps[f"enc.down{k}"] = h
            z2, zs = S.encode_image(recon0, disable_track_bn_stats=True)
            taps["enc.z_i"] = z2
            taps["enc.z_s"] = zs
            lg = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs, disable_track_bn_stats=True)
            taps["seg.logits"] = lg
            for n, t in taps.items():
                # small tensors whole, large ones as strided samples + moments
                flat = t.reshape(-1)
                res[f"tap.{n}.stats"] = np.array([float(flat.mean()), float(flat.std()), float(flat.abs().max())])
                idx = torch.linspace(0, flat.numel() - 1, 2048).long()
                res[f"tap.{n}.sample"] = flat[idx].numpy()
    return res


def main():
    MaxStyle = ref_harness.load_maxstyle_cls()
    solver_mod = ref_harness.load_solver_module()
    out_dir = HERE

    np.savez_compressed(os.path.join(out_dir, "kat_ramp.npz"), **known_answer_ramp(MaxStyle))

    layer = {}
    for tag, kw in {
        "a": dict(B=4, C=3, H=8, W=8, seed=11, dtype=torch.float32),
        "b_outside": dict(B=6, C=5, H=9, W=7, seed=12, dtype=torch.float32, lmda_outside=True),
        "c_nomix": dict(B=4, C=2, H=16, W=16, seed=13, dtype=torch.float32, mix_style=False),
        "d_f64": dict(B=4, C=3, H=8, W=8, seed=11, dtype=torch.float64),
        "e_big": dict(B=8, C=4, H=32, W=24, seed=14, dtype=torch.float32),
    }.items():
        for k, v in layer_case(MaxStyle, **kw).items():
            layer[f"{tag}.{k}"] = v
    np.savez_compressed(os.path.join(out_dir, "layer_cases.npz"), **layer)

    np.savez_compressed(os.path.join(out_dir, "mixstyle_cases.npz"), **mixstyle_cases())

    spec16 = orc.NetSpec(4, 1, 4)
    # config 1 of BASELINE.json: B=4, 1x128x128, 1 layer, 1 inner step
    c1 = loop_case(solver_mod, spec16, B=4, size=128, layers=[3], K=1, dtype=torch.float32, with_taps=True)
    np.savez_compressed(os.path.join(out_dir, "loop_c1.npz"), **c1)
    # config-2-shaped but small: 3 layers, K=5 (fp32 and its fp64 twin for tolerance calibration)
    c2s = loop_case(solver_mod, spec16, B=4, size=64, layers=[3, 4, 5], K=5, dtype=torch.float32)
    np.savez_compressed(os.path.join(out_dir, "loop_c2small.npz"), **c2s)
    c2d = loop_case(solver_mod, spec16, B=4, size=64, layers=[3, 4, 5], K=5, dtype=torch.float64)
    keep = lambda d: {k: v for k, v in d.items() if k in ("image", "losses") or k.startswith("step")}
    np.savez_compressed(os.path.join(out_dir, "loop_c2small_f64.npz"), **keep(c2d))
    c1d = loop_case(solver_mod, spec16, B=4, size=128, layers=[3], K=1, dtype=torch.float64)
    np.savez_compressed(os.path.join(out_dir, "loop_c1_f64.npz"), **keep(c1d))
    # the loop with the sub-networks in eval mode (BatchNorm running statistics), fp32 + fp64 twin
    ce = loop_case(solver_mod, spec16, B=4, size=64, layers=[3, 4, 5], K=3, dtype=torch.float32, eval_mode=True)
    np.savez_compressed(os.path.join(out_dir, "loop_eval.npz"), **ce)
    ced = loop_case(solver_mod, spec16, B=4, size=64, layers=[3, 4, 5], K=3, dtype=torch.float64, eval_mode=True)
    np.savez_compressed(os.path.join(out_dir, "loop_eval_f64.npz"), **keep(ced))
    # all six insertion points incl. layer 0 (on the code) at K=2
    c6 = loop_case(solver_mod, spec16, B=3, size=64, layers=[0, 1, 2, 3, 4, 5], K=2, dtype=torch.float32)
    np.savez_compressed(os.path.join(out_dir, "loop_all_layers.npz"), **{k: v for k, v in c6.items() if k != "z_i"})
    for f in sorted(os.listdir(out_dir)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(out_dir, f)))


if __name__ == "__main__":
    main()
