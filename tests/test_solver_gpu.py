"""API-level parity: the drop-in solver / network modules against the reference golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from parity_util import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make_solver(dev, spec_o):
    import maxstyle_amd as M
    from oracle import maxstyle_oracle as orc
    net = "FCN_16_standard_no_STN" if spec_o.reduce == 4 else "FCN_64_standard_no_STN"
    S = M.AdvancedTripletReconSegmentationModel(network_type=net, image_ch=spec_o.image_ch, num_classes=spec_o.num_classes, use_gpu=True)
    W = orc.procedural_weights(spec_o, 0)
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)       # reference state_dict key layout
        mod.train()
    return S, W


def injector(styles, dev):
    """Overwrite the random state of freshly built MaxStyle modules.  With the trainer's p = 0.5 the construction-time draw decides whether a
    module owns Parameters at all (maxstyle.py:62-73): a module that drew the other branch is re-drawn with p forced, then the state is set."""
    def hook(mod_dict):
        for k, m in mod_dict.items():
            st = styles[int(k)]
            has = "gamma_noise" in m._parameters
            if st.applied and not has:
                p0, m.p = m.p, 2.0
                m.init_parameters()
                m.p = p0
            elif not st.applied and has:
                for n in ("gamma_noise", "beta_noise", "lmda"):
                    m._parameters.pop(n, None)
                p0, m.p = m.p, -1.0
                m.init_parameters()
                m.p = p0
            m.perm = st.perm.clone(); m.rand_p = torch.tensor([0.0 if st.applied else 1.0])
            if st.applied:
                with torch.no_grad():
                    m.gamma_noise.data = st.gamma_noise.float().to(dev)
                    m.beta_noise.data = st.beta_noise.float().to(dev)
                    m.lmda.data = st.lmda.float().to(dev)
    return hook


def test_generate_max_style_image_vs_reference(golden_dir, dev):
    from oracle import maxstyle_oracle as orc
    g = np.load(os.path.join(golden_dir, "loop_c2small.npz")); g64 = np.load(os.path.join(golden_dir, "loop_c2small_f64.npz"))
    noise_img = rel(g["image"], g64["image"])                  # the reference's own fp32-vs-fp64 distance after the free-running K = 5 loop
    noise_loss = float(np.max(np.abs(g["losses"] - g64["losses"]) / np.abs(g64["losses"])))
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = injector(styles, dev)
    z_i, z_s = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    assert rel(z_i, g["z_i"]) < 3e-5
    # K = 0: plain styled decode at the injected parameters == oracle, tightly
    with torch.no_grad():
        ref0 = orc.apply_max_style(W["image_decoder"], torch.from_numpy(g["z_i"]), {i: s.clone() for i, s in styles.items()}, layers)
    out0 = S.generate_max_style_image(z_i, layers, spec.channel_num, p=1.5, n_iter=0, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    assert rel(out0, ref0) < 2e-5
    # K = 5 free-running (loose: see tests/parity_util.py), twice: the second call replays the captured HIP graph
    for rep in range(2):
        out = S.generate_max_style_image(z_i, layers, spec.channel_num, p=1.5, n_iter=5, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
        ls = S.last_losses.cpu().numpy().astype(np.float64)
        # Here the code comes from the GPU encoder (3e-5 from the reference's, asserted above), and on these RANDOMLY initialised networks the free-running loop
        # amplifies that: measured 1.5e-3 on the losses (from the reference's own code the same loop is 2e-5 / 1.1x the reference's noise from its fp64 run:
        # tests/test_engine_gpu.py::test_free_running_loop).  The bars are 3x what was measured, no longer 1e-2 / 3e-2.
        assert float(np.max(np.abs(ls - g64["losses"]) / np.abs(g64["losses"]))) < max(3.0 * noise_loss, 5e-3)
        assert rel(out, g64["image"]) < max(3.0 * noise_img, 1.5e-2), (rel(out, g64["image"]), noise_img)
        if rep == 0:
            first = out.clone()
        else:
            assert torch.equal(first, out), "second call (graph reuse) must reproduce the first bit for bit"
    # side effects the reference guarantees: requires_grad == training flag, grads cleared
    for m in S.model.values():
        assert all(p.requires_grad == m.training for p in m.parameters())
        assert all(p.grad is None for p in m.parameters())
    # Dice parity of the segmentation of the stylised image
    _, zs2 = S.encode_image(out, disable_track_bn_stats=True)
    logits = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs2, disable_track_bn_stats=True)
    np.testing.assert_allclose(orc.dice_per_class(logits.argmax(1).cpu(), lab, 4), g["final_dice"], atol=5e-3)


def test_generate_max_style_image_eval_mode(golden_dir, dev):
    """Sub-networks in .eval() when the loop is called (test-time use): BatchNorm running statistics everywhere; fixture from the reference."""
    from oracle import maxstyle_oracle as orc
    g = np.load(os.path.join(golden_dir, "loop_eval.npz"))
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    for m in S.model.values():
        m.eval()
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = injector(styles, dev)
    z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    assert rel(z_i, g["z_i"]) < 3e-5
    for rep in range(2):
        out = S.generate_max_style_image(z_i, layers, spec.channel_num, p=1.5, n_iter=3, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
        np.testing.assert_allclose(S.last_losses.cpu().numpy(), g["losses"], rtol=2e-3)
        assert rel(out, g["image"]) < 2e-2
    for m in S.model.values():
        assert not m.training and all(not p.requires_grad for p in m.parameters())     # requires_grad restored to the (False) training flag
    _, zs2 = S.encode_image(out, disable_track_bn_stats=True)
    logits = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs2, disable_track_bn_stats=True)
    np.testing.assert_allclose(orc.dice_per_class(logits.argmax(1).cpu(), lab, 4), g["final_dice"], atol=2e-2)


def test_error_behaviour_and_identity(dev):
    from oracle import maxstyle_oracle as orc
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    with pytest.raises(AssertionError, match="must provide reference"):
        S.generate_max_style_image(z_i, [3], spec.channel_num, p=1.5, n_iter=2)
    with pytest.raises(ValueError, match="not supported"):
        S.generate_max_style_image(z_i, [3], spec.channel_num, p=1.5, n_iter=2, reference_image=img.to(dev), reference_segmentation=lab.to(dev),
                                   loss_types=["foo"], loss_weights=[1])
    # every layer draws "not applied" (p <= 0): no parameters -> the plain reconstruction comes back (advanced_triplet...py:532-545)
    out = S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, p=-1.0, n_iter=5, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    plain = S.decoder_inference(decoder_name="image_decoder", latent_code=z_i, disable_track_bn_stats=True)
    assert rel(out, plain) < 1e-6
    with torch.no_grad():
        h = z_i.cpu()
        for k in range(1, 5):
            h = orc.res_up_block(W["image_decoder"], f"up{k}.", h, "Conv2")
        ref = torch.sigmoid(torch.nn.functional.conv2d(h, W["image_decoder"]["final_conv.weight"], W["image_decoder"]["final_conv.bias"]))
    assert rel(plain, ref) < 2e-5
    # empty layer list -> plain decoder inference
    out2 = S.generate_max_style_image(z_i, [], spec.channel_num)
    assert rel(out2, ref) < 2e-5
    with pytest.raises(RuntimeError, match="MI355X only"):
        S.generate_max_style_image(z_i.cpu(), [3], spec.channel_num)


def test_apply_max_style_autograd(dev):
    """MyDecoder.apply_max_style as a differentiable function of the MaxStyle parameters (user-driven loop, as the reference's)."""
    import maxstyle_amd as M
    from oracle import maxstyle_oracle as orc
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    layers = [3, 4]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 3 + i, torch.float64) for i in layers}
    mods = torch.nn.ModuleDict({str(i): M.MaxStyle(4, spec.channel_num[i], p=1.5) for i in layers})
    injector({i: s.clone(torch.float32) for i, s in styles.items()}, dev)(mods)
    out = S.model["image_decoder"].apply_max_style(z_i, mods, layers)
    target = torch.linspace(0, 1, out.numel(), device=dev).view_as(out)
    loss = ((out - target) ** 2).mean()
    loss.backward()
    W64 = {n: (t.double() if t.is_floating_point() else t) for n, t in W["image_decoder"].items()}
    st64 = {i: s.clone() for i, s in styles.items()}
    for s in st64.values():
        for nm in ("gamma_noise", "beta_noise", "lmda"):
            setattr(s, nm, getattr(s, nm).clone().requires_grad_(True))
    ref = orc.apply_max_style(W64, z_i.cpu().double(), st64, layers)
    lref = ((ref - target.cpu().double()) ** 2).mean()
    lref.backward()
    assert rel(out, ref) < 2e-5
    for i in layers:
        assert rel(mods[str(i)].gamma_noise.grad, st64[i].gamma_noise.grad) < 2e-3, i
        assert rel(mods[str(i)].beta_noise.grad, st64[i].beta_noise.grad) < 2e-3, i
        assert rel(mods[str(i)].lmda.grad, st64[i].lmda.grad) < 2e-3, i
        assert mods[str(i)].gamma_std is not None


def test_eval_mode_prediction_and_running_stats(dev):
    """Eval-mode forward (BN running statistics) for the Dice/evaluation row, and the running-statistics update of a tracking forward."""
    from oracle import maxstyle_oracle as orc
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    # a tracking train-mode forward moves running_mean/var like nn.BatchNorm2d(momentum=0.1)
    enc = S.model["image_encoder"]
    z_i, z_s = enc(img.to(dev))
    ref_bn = torch.nn.BatchNorm2d(16)
    ref_bn.train()
    with torch.no_grad():
        u = torch.nn.functional.conv2d(img, W["image_encoder"]["general_encoder.inc.0.weight"], W["image_encoder"]["general_encoder.inc.0.bias"], padding=1)
        ref_bn.weight.copy_(W["image_encoder"]["general_encoder.inc.1.weight"]); ref_bn.bias.copy_(W["image_encoder"]["general_encoder.inc.1.bias"])
        ref_bn(u)
    assert rel(enc.general_encoder.inc[1].running_mean, ref_bn.running_mean) < 1e-4
    assert rel(enc.general_encoder.inc[1].running_var, ref_bn.running_var) < 1e-4
    assert int(enc.general_encoder.inc[1].num_batches_tracked) == 1
    # eval-mode prediction == oracle with running statistics
    sd = {k: {n: v.detach().cpu() for n, v in m.state_dict().items()} for k, m in S.model.items()}
    logits = S.predict(img.to(dev))
    with torch.no_grad():
        _, zs = orc.encoder_forward(sd["image_encoder"], img, bn_mode="running")
        ref = orc.decoder_forward(sd["segmentation_decoder"], zs, "NN", bn_mode="running")
    assert rel(logits, ref) < 1e-4
    assert all(not m.training for m in S.model.values())      # like the reference, predict() leaves the solver in eval mode (advanced_triplet...py:679)


def test_run_predict_evaluate_api(dev):
    """run / predict / evaluate with the reference's signatures (advanced_triplet...py:310-328, 673-691, 914-934): eval-mode logits equal the
    oracle with running statistics, evaluate() accumulates the confusion matrix of the argmax, predict leaves the solver in eval mode."""
    from oracle import maxstyle_oracle as orc
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 321)
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], img, "running")
        ref = orc.decoder_forward(W["segmentation_decoder"], zs, "NN", None, "running")
    S.train()
    pred = S.predict(img.to(dev) * 3.0 + 1.0)              # normalize_input=True: per-sample min-max brings it back to [0, 1]
    assert not S.training and all(not m.training for m in S.model.values())
    assert rel(pred, ref) < 1e-4
    prob = S.predict(img.to(dev), softmax=True)
    assert rel(prob, torch.softmax(ref, 1)) < 1e-4
    S.running_metric = S.set_running_metric()
    out = S.evaluate(img.to(dev), lab.numpy())
    cm = S.running_metric.confusion_matrix().cpu()
    pl = ref.argmax(1)
    ref_cm = torch.zeros(4, 4, dtype=torch.int64)
    for t in range(4):
        for q in range(4):
            ref_cm[t, q] = int(((lab == t) & (pl == q)).sum())
    assert int((cm - ref_cm).abs().sum()) <= 4              # an argmax tie at fp32 rounding may move a pixel
    assert S.cur_eval_predicts.shape == (4, 64, 64) and S.cur_eval_gts.shape == (4, 64, 64)
    S.train()
    recon, p0, p1 = S.run(img.to(dev))                     # train mode, tracking: batch statistics (this updates the running buffers)
    assert p0 is p1 and recon.shape == img.shape and S.z_i is not None
    S.train(if_testing=True)
    assert not S.training


def test_gpu_dice_and_confusion(dev):
    """HIP confusion matrix / Dice == the oracle's label-based Dice (medpy.metric.binary.dc semantics)."""
    from maxstyle_amd.metrics import runningScore
    from oracle import maxstyle_oracle as orc
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5, 4, 48, 40, generator=g)
    lab = torch.randint(0, 4, (5, 48, 40), generator=g)
    rs = runningScore(4, dev)
    rs.update(lab.to(dev), logits.to(dev))
    pred = logits.argmax(1)
    cm = torch.zeros(4, 4, dtype=torch.int64)
    for t, p_ in zip(lab.reshape(-1), pred.reshape(-1)):
        cm[t, p_] += 1
    assert torch.equal(rs.confusion_matrix().cpu(), cm)
    np.testing.assert_allclose(rs.dice(), orc.dice_per_class(pred, lab, 4), atol=1e-12)
    scores, iou = rs.get_scores()
    assert abs(scores['Overall Acc: \t'] - float((pred == lab).double().mean())) < 1e-12
    rs.update(lab.to(dev), logits.to(dev))                     # accumulates
    assert int(rs.confusion_matrix().sum()) == 2 * lab.numel()


def test_weight_update_keeps_graph_and_matches_fresh_pack(dev):
    """After an (emulated) optimiser step the packed weights are refreshed in place: the captured graph is reused and the result equals
    a freshly built solver with the new weights, bit for bit."""
    from oracle import maxstyle_oracle as orc
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = injector(styles, dev)
    z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    kw = dict(p=1.5, n_iter=3, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    S.generate_max_style_image(z_i, layers, spec.channel_num, **kw)
    eng = next(iter(S._engines.values()))
    graph = eng._graph
    assert graph is not None
    with torch.no_grad():                                   # "optimiser step": perturb every parameter
        for m in S.model.values():
            for p_ in m.parameters():
                p_.mul_(1.01).add_(0.001)
    out = S.generate_max_style_image(z_i, layers, spec.channel_num, **kw)
    assert eng._graph is graph, "graph must survive a weight update"
    S2, _ = make_solver(dev, spec)
    for name in S.model:
        S2.model[name].load_state_dict(S.model[name].state_dict())
    S2.style_init_hook = injector(styles, dev)
    ref = S2.generate_max_style_image(z_i, layers, spec.channel_num, **kw)
    assert torch.equal(out, ref)


def test_ragged_shapes_one_step_vs_oracle(dev):
    """Non-square image, odd batch, layers [2,4]: one loss/gradient evaluation vs the fp64 oracle."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_engine_gpu import build_engine
    from oracle import maxstyle_oracle as orc
    from maxstyle_amd import engine as E
    spec = orc.NetSpec(4, 1, 4)
    B, H, Wd, layers = 5, 48, 80, [2, 4]
    W = orc.procedural_weights(spec, 0)
    to = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    eng = E.InnerLoopEngine(E.NetSpec(4, 1, 4), B, H, Wd, dev)
    eng.set_nets(E.PackedNets(E.NetSpec(4, 1, 4), to(W["image_encoder"]), to(W["segmentation_decoder"]), to(W["image_decoder"])))
    g = torch.Generator().manual_seed(3)
    img = torch.rand(B, 1, H, Wd, generator=g)
    lab = torch.randint(0, 4, (B, H, Wd), generator=g)
    styles = {i: orc.random_style_state(B, spec.channel_num[i], 11 + i) for i in layers}
    eng.configure_styles(layers, {i: E.StyleSlot(i, B, spec.channel_num[i]) for i in layers})
    for i in layers:
        st = styles[i]
        eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    W64 = {k: {n: (t.double() if t.is_floating_point() else t) for n, t in sd.items()} for k, sd in W.items()}
    with torch.no_grad():
        z_i = orc.encoder_forward(W64["image_encoder"], img.double())[0]
    recon, loss, grads = orc.inner_step_grads(W64, z_i, {i: s.clone(torch.float64) for i, s in styles.items()}, layers, lab)
    eng.code = z_i.float().to(dev)
    img_g, loss_g = eng.step_grads(lab.to(dev))
    assert rel(img_g, recon) < 1e-4
    assert abs(float(loss_g) - loss) < 1e-4 * abs(loss)
    for n, gr in grads.items():
        i, nm = n.split(".")
        assert rel(eng.grad(int(i), nm), gr) < 0.1, n


def test_config_block_and_rescale(dev):
    """The reference's JSON `max_style` block drives the call verbatim; rescale_intensity == the reference formula."""
    from maxstyle_amd import ops
    from oracle import maxstyle_oracle as orc
    cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 2, "mix_learnable": True, "noise_learnable": True,
           "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 1234)
    z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    torch.manual_seed(3)
    out = S.generate_max_style_image_from_config(z_i, cfg, img.to(dev), lab.to(dev), p=0.5, rescale=True)
    assert out.shape == (4, 1, 64, 64)
    mn = out.view(4, -1).min(1).values; mx = out.view(4, -1).max(1).values
    assert float(mn.abs().max()) == 0.0 and float((mx - 1).abs().max()) < 1e-6
    x = torch.randn(3, 2, 17, 9, device=dev) * 3 + 1
    flat = x.view(6, -1)
    ref = ((flat - flat.min(1, keepdim=True).values) / (flat.max(1, keepdim=True).values - flat.min(1, keepdim=True).values + 1e-20)).view_as(x)
    assert rel(ops.rescale_intensity(x), ref) < 1e-6
