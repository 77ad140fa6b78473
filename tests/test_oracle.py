"""The CPU oracle (oracle/maxstyle_oracle.py) pinned against the reference's golden vectors.

Fixtures come from tests/golden/make_golden.py, which imports the reference in the build container.
Tolerances: fp32 single-step quantities 1e-5 relative-to-max (same arithmetic, different op order);
K=5 trajectories are compared against the reference's own fp32-vs-fp64 noise (SURVEY 7 'Hard parts')."""
import os

import numpy as np
import pytest
import torch

from oracle import maxstyle_oracle as orc

torch.set_num_threads(1)


from parity_util import rel, ref_params_at, style_names


def test_known_answer_ramp(golden_dir):
    """The reference's only known-answer material (maxstyle.py:193-241)."""
    g = np.load(os.path.join(golden_dir, "kat_ramp.npz"))
    # the values quoted in SURVEY.md 8(c), so the fixture itself is pinned to the survey's probe
    np.testing.assert_array_equal(g["perm"], [0, 1, 3, 2])
    np.testing.assert_allclose(g["losses"], [4876.3818, 4312.5464, 3789.7957, 3308.5969, 2869.0017], rtol=1e-6)
    np.testing.assert_allclose(g["beta_std"].ravel(), [30.983868, 30.983868], rtol=1e-6)
    np.testing.assert_allclose(g["outputs"][0].ravel()[:5], [30.986458, 33.986458, 36.986458, 39.986458, -15.377647], rtol=1e-6)
    x = torch.from_numpy(g["features"])
    st = orc.StyleState(perm=torch.from_numpy(g["perm"]), lmda=torch.from_numpy(g["lmda0"]).clone(),
                        gamma_noise=torch.from_numpy(g["gamma_noise0"]).clone(), beta_noise=torch.from_numpy(g["beta_noise0"]).clone())
    params = [st.gamma_noise.requires_grad_(True), st.beta_noise.requires_grad_(True), st.lmda.requires_grad_(True)]
    m = [torch.zeros_like(p) for p in params]; v = [torch.zeros_like(p) for p in params]
    for i in range(5):
        y = orc.maxstyle_forward(x, st)
        loss = ((y - 1) ** 2).mean()
        grads = torch.autograd.grad(loss, params)
        assert rel(y.detach().numpy(), g["outputs"][i]) < 1e-5
        assert abs(float(loss) - g["losses"][i]) / g["losses"][i] < 1e-5
        with torch.no_grad():
            for p, gr, mm, vv in zip(params, grads, m, v):
                orc.adam_step(p, gr, mm, vv, i + 1, 0.1)
    assert rel(st.gamma_noise.detach().numpy(), g["gamma_noise5"]) < 1e-4
    assert rel(st.beta_noise.detach().numpy(), g["beta_noise5"]) < 1e-4
    assert rel(st.lmda.detach().numpy(), g["lmda5"]) < 1e-4
    np.testing.assert_allclose(st.gamma_std.numpy(), g["gamma_std"], atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b_outside", "c_nomix", "d_f64", "e_big"])
def test_layer_forward_backward(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "layer_cases.npz"))
    t = lambda k: torch.from_numpy(g[f"{tag}.{k}"])
    x, dy = t("x"), t("dy")
    st = orc.StyleState(perm=t("perm"), lmda=t("lmda"), gamma_noise=t("gamma_noise"), beta_noise=t("beta_noise"),
                        mix_style=(tag != "c_nomix"))
    y, mu, sig = orc.maxstyle_forward(x, st, return_stats=True)
    tol = 1e-12 if tag == "d_f64" else 2e-6
    assert rel(y, g[f"{tag}.y"]) < tol
    assert rel(st.gamma_std, g[f"{tag}.gamma_std"]) < max(tol, 1e-6)
    assert rel(st.beta_std, g[f"{tag}.beta_std"]) < max(tol, 1e-6)
    dx, dg, db, dl = orc.maxstyle_backward(dy, x, mu, sig, st)
    gtol = 1e-11 if tag == "d_f64" else 2e-5
    assert rel(dx, g[f"{tag}.dx"]) < gtol
    assert rel(dg, g[f"{tag}.d_gamma"]) < gtol
    assert rel(db, g[f"{tag}.d_beta"]) < gtol
    if tag != "c_nomix":
        assert rel(dl, g[f"{tag}.d_lmda"]) < gtol
        if tag == "b_outside":  # lmda outside [0,1]: gradient exactly zero (clamp)
            assert float(dl[0]) == 0.0 and float(dl[-1]) == 0.0


def test_identity_paths():
    x = torch.randn(4, 3, 5, 5)
    st = orc.random_style_state(4, 3, 1, applied=False)
    assert orc.maxstyle_forward(x, st) is x                     # rand_p >= p
    st = orc.random_style_state(4, 3, 1); st.mix_style = False; st.no_noise = True
    assert orc.maxstyle_forward(x, st) is x                     # nothing to do
    x1 = torch.randn(4, 3, 1, 1)
    assert orc.maxstyle_forward(x1, orc.random_style_state(4, 3, 1)) is x1   # H*W == 1
    xb = torch.randn(1, 3, 5, 5)
    assert orc.maxstyle_forward(xb, orc.random_style_state(1, 3, 1)) is xb   # B <= 1


def _setup(spec, B, size, layers, dtype=torch.float32, style_seed=7):
    W = orc.procedural_weights(spec, 0, dtype)
    img, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, 1234)
    img = img.to(dtype)
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img)
    styles = {i: orc.random_style_state(B, spec.channel_num[i], style_seed + i, dtype) for i in layers}
    return W, img, lab, z_i, styles


def _teacher_forced(g, g64, spec, B, size, layers, K, grad_tol_later=5e-2, bn_mode="batch"):
    """Every step from the reference's own parameters: loss/forward tight, grads noise-calibrated."""
    W, img, lab, z_i, styles = _setup(spec, B, size, layers)
    if bn_mode != "batch":
        with torch.no_grad():
            z_i, _ = orc.encoder_forward(W["image_encoder"], img, bn_mode)
        assert rel(z_i, g["z_i"]) < 1e-5
    initial = {f"{i}.{n}": getattr(styles[i], n).numpy().copy() for i in layers for n in ("gamma_noise", "beta_noise", "lmda")}
    for s in range(1, K + 1):
        cur = ref_params_at(g, s - 1, layers, initial)
        for n, val in cur.items():
            i, nm = n.split(".")
            setattr(styles[int(i)], nm, torch.from_numpy(np.array(val)).clone())
        recon, loss, grads = orc.inner_step_grads(W, z_i, styles, layers, lab, bn_mode)
        assert abs(loss - g["losses"][s - 1]) < 2e-5 * abs(g["losses"][s - 1]), (s, loss, g["losses"][s - 1])
        for n in style_names(layers):
            ref = g[f"step{s}.grad.{n}"]
            if s == 1 and g64 is not None:
                noise = rel(ref, g64[f"step1.grad.{n}"])
                assert rel(grads[n], g64[f"step1.grad.{n}"]) < max(4 * noise, 1e-4), (s, n, noise)
            else:
                assert rel(grads[n], ref) < grad_tol_later, (s, n)
        if s == 1:
            for i in layers:
                assert rel(styles[i].gamma_std, g[f"{i}.gamma_std"]) < 1e-5
                assert rel(styles[i].beta_std, g[f"{i}.beta_std"]) < 1e-5
        # Adam given the reference's gradient reproduces the reference's parameters
        if s == 1:
            for n in style_names(layers):
                p = torch.from_numpy(np.array(cur[n])).clone()
                gr = torch.from_numpy(np.array(g[f"step1.grad.{n}"]))
                orc.adam_step(p, gr, torch.zeros_like(p), torch.zeros_like(p), 1, 0.1)
                assert rel(p, g[f"step1.param.{n}"]) < 1e-6, n
    # final decode from the reference's final parameters == the reference's returned image
    for n, val in ref_params_at(g, K, layers, initial).items():
        i, nm = n.split(".")
        setattr(styles[int(i)], nm, torch.from_numpy(np.array(val)).clone())
    with torch.no_grad():
        out = orc.apply_max_style(W["image_decoder"], z_i, styles, layers, bn_mode=bn_mode)
    assert rel(out, g["image"]) < 1e-5
    return W, lab, out


def test_loop_config1(golden_dir):
    """BASELINE config 1: B=4, 1x128x128, one MaxStyle layer, one inner step."""
    g = np.load(os.path.join(golden_dir, "loop_c1.npz"))
    g64 = np.load(os.path.join(golden_dir, "loop_c1_f64.npz"))
    spec = orc.NetSpec(4, 1, 4)
    W, lab, out = _teacher_forced(g, g64, spec, 4, 128, [3], 1)
    # block-level activations of the un-styled decoder/encoder/seg decoder (BN batch-stat mode)
    taps = {}
    with torch.no_grad():
        h = torch.from_numpy(g["z_i"])
        for k in range(1, 5):
            h = orc.res_up_block(W["image_decoder"], f"up{k}.", h, "Conv2")
            taps[f"dec.up{k}"] = h
        recon = torch.sigmoid(torch.nn.functional.conv2d(h, W["image_decoder"]["final_conv.weight"], W["image_decoder"]["final_conv.bias"]))
        taps["dec.recon"] = recon
        et = {}
        zi, zs = orc.encoder_forward(W["image_encoder"], recon, taps=et)
        taps["enc.inc"] = et["general_encoder.inc.out"]
        for k in range(1, 5):
            taps[f"enc.down{k}"] = et[f"general_encoder.down{k}.out"]
        taps["enc.z_i"], taps["enc.z_s"] = zi, zs
        taps["seg.logits"] = orc.decoder_forward(W["segmentation_decoder"], zs, "NN")
    for n, t in taps.items():
        flat = t.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 2048).long()
        assert rel(flat[idx], g[f"tap.{n}.sample"]) < 2e-5, n


def test_loop_k5_three_layers(golden_dir):
    """Config-2-shaped (layers [3,4,5], K=5) at 4x1x64x64: per-step losses/grads, final image, segmentation, Dice."""
    g = np.load(os.path.join(golden_dir, "loop_c2small.npz"))
    g64 = np.load(os.path.join(golden_dir, "loop_c2small_f64.npz"))
    spec = orc.NetSpec(4, 1, 4)
    W, lab, out = _teacher_forced(g, g64, spec, 4, 64, [3, 4, 5], 5)
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], out)
        pred = orc.decoder_forward(W["segmentation_decoder"], zs, "NN").argmax(1)
    assert float((pred.numpy() == g["final_pred"]).mean()) > 0.9995
    np.testing.assert_allclose(orc.dice_per_class(pred, lab, 4), g["final_dice"], atol=1e-3)


def test_loop_config4_shaped_network(golden_dir):
    """BASELINE config 4's network - FCN_64 widths, 3 image channels, 2 classes - against the reference's OWN run of it (loop_c4small.npz, 4x3x64x64,
    layers [3,4,5], K=3): per-step losses / gradients / parameters teacher-forced, final image, segmentation, Dice; fp64 restatement == fp64 reference."""
    g = np.load(os.path.join(golden_dir, "loop_c4small.npz"))
    g64 = np.load(os.path.join(golden_dir, "loop_c4small_f64.npz"))
    spec = orc.NetSpec(1, 3, 2)
    W, lab, out = _teacher_forced(g, g64, spec, 4, 64, [3, 4, 5], 3)
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], out)
        pred = orc.decoder_forward(W["segmentation_decoder"], zs, "NN").argmax(1)
    assert float((pred.numpy() == g["final_pred"]).mean()) > 0.9995
    np.testing.assert_allclose(orc.dice_per_class(pred, lab, 2), g["final_dice"], atol=1e-3)
    W, img, lab, z_i, styles = _setup(spec, 4, 64, [3, 4, 5], dtype=torch.float64)
    tr = orc.InnerLoopTrace()
    out64 = orc.generate_max_style_image(W, z_i, styles, [3, 4, 5], lab, n_iter=3, lr=0.1, trace=tr)
    assert rel(out64, g64["image"]) < 1e-9
    np.testing.assert_allclose(tr.losses, g64["losses"], rtol=1e-10)


def test_loop_eval_mode(golden_dir):
    """The loop called with the sub-networks in .eval() (running statistics in every BatchNorm): K=3, layers [3,4,5]."""
    g = np.load(os.path.join(golden_dir, "loop_eval.npz"))
    g64 = np.load(os.path.join(golden_dir, "loop_eval_f64.npz"))
    spec = orc.NetSpec(4, 1, 4)
    # fp32 gradients against the reference's fp32 gradients (activation-mask flips: see parity_util); exactness is pinned by the fp64 twin below
    W, lab, out = _teacher_forced(g, None, spec, 4, 64, [3, 4, 5], 3, grad_tol_later=1e-2, bn_mode="running")
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], out, "running")
        pred = orc.decoder_forward(W["segmentation_decoder"], zs, "NN", None, "running").argmax(1)
    assert float((pred.numpy() == g["final_pred"]).mean()) > 0.9995
    np.testing.assert_allclose(orc.dice_per_class(pred, lab, 4), g["final_dice"], atol=1e-3)
    # fp64 restatement == fp64 reference to rounding
    W, img, lab, z_i, styles = _setup(spec, 4, 64, [3, 4, 5], dtype=torch.float64)
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img, "running")
    out64 = orc.generate_max_style_image(W, z_i, styles, [3, 4, 5], lab, n_iter=3, lr=0.1, bn_mode="running")
    assert rel(out64, g64["image"]) < 1e-9


def test_loop_free_running(golden_dir):
    """Un-forced K=5 trajectory: only loosely comparable (see tests/parity_util.py), Dice must still agree."""
    g = np.load(os.path.join(golden_dir, "loop_c2small.npz"))
    spec = orc.NetSpec(4, 1, 4)
    W, img, lab, z_i, styles = _setup(spec, 4, 64, [3, 4, 5])
    tr = orc.InnerLoopTrace()
    out = orc.generate_max_style_image(W, z_i, styles, [3, 4, 5], lab, n_iter=5, lr=0.1, trace=tr)
    np.testing.assert_allclose(tr.losses, g["losses"], rtol=5e-3)
    assert rel(out, g["image"]) < 2e-2
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], out)
        pred = orc.decoder_forward(W["segmentation_decoder"], zs, "NN").argmax(1)
    np.testing.assert_allclose(orc.dice_per_class(pred, lab, 4), g["final_dice"], atol=2e-2)


def test_loop_all_six_layers(golden_dir):
    g = np.load(os.path.join(golden_dir, "loop_all_layers.npz"))
    _teacher_forced(g, None, orc.NetSpec(4, 1, 4), 3, 64, [0, 1, 2, 3, 4, 5], 2, grad_tol_later=0.2)  # B=3, 4x4 code: very noisy fp32 grads


def test_oracle_fp64_matches_reference_fp64(golden_dir):
    g64 = np.load(os.path.join(golden_dir, "loop_c2small_f64.npz"))
    spec = orc.NetSpec(4, 1, 4)
    W, img, lab, z_i, styles = _setup(spec, 4, 64, [3, 4, 5], dtype=torch.float64)
    tr = orc.InnerLoopTrace()
    out = orc.generate_max_style_image(W, z_i, styles, [3, 4, 5], lab, n_iter=5, lr=0.1, trace=tr)
    # same algorithm, same inputs, fp64 on both sides: agreement to rounding pins the restatement exactly
    assert rel(out, g64["image"]) < 1e-9
    np.testing.assert_allclose(tr.losses, g64["losses"], rtol=1e-10)
    for k in g64.files:
        if k.startswith("step5.param."):
            assert rel(tr.params[4][k[len("step5.param."):]], g64[k]) < 1e-8, k
        if k.startswith("step1.grad."):
            assert rel(tr.grads[0][k[len("step1.grad."):]], g64[k]) < 1e-8, k


@pytest.mark.parametrize("tag", ["random", "cross", "extrap", "dsu"])
def test_mixstyle_restatement_vs_reference(golden_dir, tag):
    """oracle.mixstyle_forward (MixStyle / DSU with injected draws) against the reference's own outputs and input gradients (mixstyle_cases.npz)."""
    g = np.load(os.path.join(golden_dir, "mixstyle_cases.npz"))
    x = torch.from_numpy(g[f"{tag}.x"]).requires_grad_(True)
    if tag == "dsu":
        y = orc.mixstyle_forward(x, gaussian_mu=torch.from_numpy(g["dsu.gaussian_mu"]), gaussian_std=torch.from_numpy(g["dsu.gaussian_std"]))
    else:
        lm = torch.from_numpy(g[f"{tag}.lmda"]) if f"{tag}.lmda" in g.files else torch.full((x.shape[0], 1, 1, 1), 1.7)
        y = orc.mixstyle_forward(x, perm=torch.from_numpy(g[f"{tag}.perm"]), lmda=lm)
    y.backward(torch.from_numpy(g[f"{tag}.dy"]))
    assert rel(y, g[f"{tag}.y"]) < 1e-6
    assert rel(x.grad, g[f"{tag}.dx"]) < 1e-6


# ---- round 2: random-depth insertion (p = 0.5 literal of the trainer) and the loop on trained networks ------------------------------------
@pytest.mark.parametrize("tag", ["only3", "l45", "none"])
def test_loop_random_depth_vs_reference(golden_dir, tag):
    """The oracle's not-applied path (identity, no parameters, no Adam state) against the reference run with injected rand_p
    (tests/golden/make_golden_r2.py: strict subsets {3}, {4,5}, {} of the inserted layers [3,4,5])."""
    g = np.load(os.path.join(golden_dir, "loop_random_depth.npz"))
    applied = set(int(i) for i in g[f"{tag}.applied"])
    spec = orc.NetSpec(4, 1, 4)
    W = orc.procedural_weights(spec, 0)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i, applied=(i in applied)) for i in layers}
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img)
    tr = orc.InnerLoopTrace()
    out = orc.generate_max_style_image(W, z_i, styles, layers, lab, n_iter=2, lr=0.1, trace=tr)
    ref_losses = g[f"{tag}.losses"]
    assert len(tr.losses) == len(ref_losses)
    if applied:
        assert abs(tr.losses[0] - ref_losses[0]) < 1e-5 * abs(ref_losses[0])
        np.testing.assert_allclose(tr.losses, ref_losses, rtol=5e-3)
        assert rel(out, g[f"{tag}.image"]) < 2e-2          # free-running after one lr*sign(g) step: loose (parity_util)
    else:
        assert rel(out, g[f"{tag}.image"]) < 1e-5


def _trained_weights(golden_dir, dtype):
    z = np.load(os.path.join(golden_dir, "trained_fcn16.npz"))
    W = {"image_encoder": {}, "segmentation_decoder": {}, "image_decoder": {}}
    for key in z.files:
        net, name = key.split("/", 1)
        a = z[key]
        t = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
        W[net][name] = t.to(dtype) if t.is_floating_point() else t
    return W


def test_loop_on_trained_networks_fp64_matches_reference(golden_dir):
    """Networks trained by the reference's own training step (fixture trained_fcn16.npz): the fp64 oracle reproduces the reference's fp64 K=5 loop,
    its Dice values and its predictions; the fixture's Dice is meaningful (clean >= 0.65 per class, stylised lower)."""
    g = np.load(os.path.join(golden_dir, "loop_trained.npz"))
    W = _trained_weights(golden_dir, torch.float64)
    spec = orc.NetSpec(4, 1, 4)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 777)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i, torch.float64) for i in layers}
    with torch.no_grad():
        z_i, z_s = orc.encoder_forward(W["image_encoder"], img.double())
        clean_pred = orc.decoder_forward(W["segmentation_decoder"], z_s, "NN").argmax(1)
    assert rel(z_i, g["f64.z_i"]) < 1e-10
    np.testing.assert_allclose(orc.dice_per_class(clean_pred, lab, 4), g["f64.clean_dice"], atol=1e-12)
    tr = orc.InnerLoopTrace()
    out = orc.generate_max_style_image(W, z_i, styles, layers, lab, n_iter=5, lr=0.1, trace=tr)
    np.testing.assert_allclose(tr.losses, g["f64.losses"], rtol=1e-8)
    assert rel(out, g["f64.image"]) < 1e-7
    with torch.no_grad():
        _, zs2 = orc.encoder_forward(W["image_encoder"], out)
        pred = orc.decoder_forward(W["segmentation_decoder"], zs2, "NN").argmax(1)
    np.testing.assert_allclose(orc.dice_per_class(pred, lab, 4), g["f64.final_dice"], atol=1e-12)
    assert min(g["f64.clean_dice"]) > 0.6 and max(g["f64.final_dice"]) < min(g["f64.clean_dice"])
