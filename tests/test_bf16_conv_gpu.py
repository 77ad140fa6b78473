"""bf16 ACTIVATION STORAGE of the conv stack (BASELINE config 5; include/maxstyle_hip.h `_bf16` entry points): every op equals its fp32 twin evaluated on
the bf16-rounded inputs, with the output rounded to nearest-even bf16 - i.e. |out_bf16 - round_bf16(ref64)| <= one bf16 ulp of the value where the
fp32 arithmetic lands on a rounding boundary, and 2^-8 relative everywhere; statistics / coefficient tables (fp32) agree with the fp32 op to 1e-5.
Tolerances are stated per test.  The arithmetic (fp32 matrix cores, fp32 statistics) is the fp32 path's: only loads and stores differ."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rb(t):
    """bf16 rounding of an fp32 CPU tensor, kept as fp32 / fp64 values."""
    return t.to(BF).to(torch.float32)


def close_bf16(got_bf16, ref64, extra=0.0):
    """got (bf16 device tensor) against an fp64 reference: within 2^-8 relative of the largest magnitude in a plane-independent sense."""
    g = got_bf16.float().cpu().double()
    err = (g - ref64).abs()
    tol = ref64.abs() * 2.0 ** -8 + ref64.abs().max() * (2.0 ** -13 + extra)
    bad = err > tol
    assert not bool(bad.any()), (int(bad.sum()), float(err.max()), float(ref64.abs().max()))


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks,stride", [
    (2, 16, 16, 64, 64, 3, 1), (1, 32, 48, 20, 192, 3, 1), (2, 64, 64, 64, 64, 3, 1), (1, 8, 33, 8, 100, 3, 1),      # wide-read kernel: NT 1 / 2, ragged tiles, channel tails
    (2, 16, 16, 32, 32, 3, 1), (2, 128, 128, 16, 16, 3, 1), (1, 20, 24, 9, 72, 3, 1), (2, 24, 40, 6, 12, 3, 1),      # first-generation kernel: wide and narrow tiles
    (2, 32, 16, 12, 20, 1, 1), (2, 16, 32, 32, 32, 3, 2), (2, 1, 16, 32, 32, 3, 1),                                   # 1x1, stride 2, the image-input layer
])
def test_conv2d_bf16_all_prologues(dev, N, Cin, Cout, H, W, ks, stride):
    from maxstyle_amd import ops
    x = rb(_rand((N, Cin, H, W), 1)); x2 = rb(_rand((N, Cin, H, W), 2)); w = _rand((Cout, Cin, ks, ks), 3, 0.1); b = _rand((Cout,), 4)
    cf = _rand((Cin, 4), 5)
    wp = ops.pack_conv_weight(w.to(dev))
    pad = ks // 2
    a, bb, cc = cf[:, 0].double().view(1, -1, 1, 1), cf[:, 1].double().view(1, -1, 1, 1), cf[:, 2].double().view(1, -1, 1, 1)
    cfd = cf.to(dev)
    # plain + bias, with the BatchNorm statistics epilogue
    Ho, Wo = ops.conv_out_hw(H, W, ks, stride, 0)
    stats, parts = ops.conv_stats_buffer(N, Cout, Ho, Wo, dev)
    out = ops.conv2d(x.to(dev).to(BF), wp, b.to(dev), Cout, ks, stride, stats=stats)
    assert out.dtype == BF
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    close_bf16(out, ref)
    gamma = torch.ones(Cout, device=dev); beta = torch.zeros(Cout, device=dev)
    coef = ops.bn_finalize(stats, parts, gamma, beta).cpu().double()
    mean = ref.mean((0, 2, 3)); invstd = 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)
    assert float((coef[:, 2] - mean).abs().max()) < 1e-5 * max(1.0, float(mean.abs().max()))           # statistics of the fp32 accumulators: fp32 accuracy
    assert float((coef[:, 3] / invstd - 1).abs().max()) < 1e-5
    if stride == 1:
        # BatchNorm apply + LeakyReLU prologue
        o1 = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, ks, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
        close_bf16(o1, F.conv2d(F.leaky_relu(a * x.double() + bb, 0.2), w.double(), None, padding=pad))
        # two-tensor BatchNorm-backward prologue, accumulate epilogue
        base = rb(_rand((N, Cout, H, W), 6))
        o2 = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, ks, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                        pro_cstride=4, in2=x2.to(dev).to(BF), epi_mode=1, out=base.to(dev).to(BF).clone())
        close_bf16(o2, F.conv2d(a * x.double() + bb * x2.double() + cc, w.double(), None, padding=pad) + base.double())


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 128, 128, 16, 16), (2, 16, 16, 32, 32)])
def test_conv2d_actbwd_bf16(dev, N, Cin, Cout, H, W):
    """Activation-backward epilogue on bf16 storage: masked gradient against fp64 math on the rounded inputs (away from the activation's kink), the
    BatchNorm-backward coefficient table against the fp32 reduce of the SAME masked gradient computed in fp64."""
    from maxstyle_amd import ops
    g = rb(_rand((N, Cin, H, W), 21)); w = _rand((Cout, Cin, 3, 3), 23, 0.1)
    u = rb(_rand((N, Cout, H, W), 24) + 0.3)
    coef = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1)
    wp = ops.pack_conv_weight(w.to(dev))
    out, tab = ops.conv2d_actbwd(g.to(dev).to(BF), wp, Cout, 3, u.to(dev).to(BF), coef.to(dev), 0.2)
    assert out.dtype == BF
    cc = coef.double()
    ref = F.conv2d(g.double(), w.double(), None, padding=1)
    pre = cc[:, 0].view(1, -1, 1, 1) * u.double() + cc[:, 1].view(1, -1, 1, 1)
    safe = (pre.abs() > 1e-4).double()
    refm = ref * torch.where(pre > 0, 1.0, 0.2)
    close_bf16((out.float() * safe.float().to(dev)).to(BF), refm * safe)
    bc = ops.bn_bwd_coefs(tab, 0, coef.to(dev), N * H * W).cpu().double()
    s1 = refm.sum((0, 2, 3)); s2 = (refm * (u.double() - cc[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
    cnt = N * H * W
    be = -cc[:, 0] * (s2 * cc[:, 3] / cnt) * cc[:, 3]
    ref_bc = torch.stack([cc[:, 0], be, -cc[:, 0] * s1 / cnt - be * cc[:, 2]], 1)
    assert float((bc[:, :3] - ref_bc).abs().max()) < 2e-4 * float(ref_bc.abs().max())      # the sums are taken from the fp32 values before rounding


def test_streaming_kernels_bf16(dev):
    """bn_act (all residual modes), act_bwd_reduce (both mask sources), pool2_sum, the sigmoid head and its backward, the cross-entropy head: bf16 twins."""
    from maxstyle_amd import ops
    N, C, H, W = 2, 16, 32, 48
    u = rb(_rand((N, C, H, W), 1)); res = rb(_rand((N, C, H, W), 2)); half = rb(_rand((N, C, H // 2, W // 2), 3))
    coef = torch.stack([1 + 0.2 * _rand((C,), 4), 0.3 * _rand((C,), 5), 0.1 * _rand((C,), 6), 1 + 0.1 * _rand((C,), 7).abs()], 1)
    cd = coef.to(dev); c64 = coef.double()
    sc, sh = c64[:, 0].view(1, -1, 1, 1), c64[:, 1].view(1, -1, 1, 1)
    ub = u.to(dev).to(BF)
    close_bf16(ops.bn_act(ub, cd, slope=0.2), F.leaky_relu(sc * u.double() + sh, 0.2))
    close_bf16(ops.bn_act(ub, cd, res.to(dev).to(BF), 1, slope=0.2), F.leaky_relu(sc * u.double() + sh + res.double(), 0.2))
    close_bf16(ops.bn_act(ub, cd, half.to(dev).to(BF), 2, slope=0.0), F.relu(sc * u.double() + sh + F.interpolate(half.double(), scale_factor=2, mode="nearest")))
    g = rb(_rand((N, C, H, W), 8))
    act = F.leaky_relu(sc * u.double() + sh, 0.2)
    for ref_t in (rb(act.float()).to(dev).to(BF), None):
        gm, part, nparts = ops.act_bwd_reduce(g.to(dev).to(BF), ref_t, ub, cd, 0.2)
        pre = sc * u.double() + sh
        mask_src = act if ref_t is not None else pre
        safe = (mask_src.abs() > 1e-3).double()
        refm = g.double() * torch.where(mask_src > 0, 1.0, 0.2)
        close_bf16((gm.float() * safe.float().to(dev)).to(BF), refm * safe)
        bc = ops.bn_bwd_coefs(part, nparts, cd, N * H * W).cpu().double()
        gm64 = gm.float().cpu().double()                                   # the sums are over the UNROUNDED masked gradient: compare with a loose bound
        s1 = gm64.sum((0, 2, 3)); s2 = (gm64 * (u.double() - c64[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
        cnt = N * H * W
        be = -c64[:, 0] * (s2 * c64[:, 3] / cnt) * c64[:, 3]
        assert float((bc[:, 1] - be).abs().max()) < 2e-2 * float(be.abs().max()) + 1e-6
    x = rb(_rand((N, C, H, W), 9))
    close_bf16(ops.pool2_sum(x.to(dev).to(BF)), F.avg_pool2d(x.double(), 2) * 4)
    hw = _rand((1, C), 10, 0.3); hb = _rand((1,), 11)
    img = ops.head_fwd(ub, hw.to(dev), hb.to(dev), True)
    ref_img = torch.sigmoid(F.conv2d(u.double(), hw.double().view(1, C, 1, 1), hb.double()))
    close_bf16(img, ref_img)
    dout = rb(_rand((N, 1, H, W), 12))
    o = img.float().cpu().double()
    dh = ops.head_bwd(dout.to(dev).to(BF), img, hw.to(dev), C, True)
    close_bf16(dh, F.conv_transpose2d(dout.double() * o * (1 - o), hw.double().view(1, C, 1, 1)))
    K = 4
    cw = _rand((K, C), 13, 0.3); cb = _rand((K,), 14)
    lab = torch.randint(0, K, (N, H, W), generator=torch.Generator().manual_seed(15))
    loss, dh2, logits = ops.head_ce(ub, cw.to(dev), cb.to(dev), lab.to(dev), loss_sign=-1.0, need_logits=True)
    h64 = u.double().requires_grad_(True)
    z = F.conv2d(h64, cw.double().view(K, C, 1, 1), cb.double())
    ce = -F.cross_entropy(z, lab)
    ce.backward()
    assert abs(float(loss) - float(ce)) < 1e-5 * abs(float(ce))
    assert logits.dtype == torch.float32 and float((logits.cpu().double() - z.detach()).abs().max()) < 1e-5 * float(z.abs().max())
    close_bf16(dh2, h64.grad)


# ------------------------------------------------------------------------------------------------ the inner loop on bf16 activation storage
def _engines(dev, spec, B, size, layers):
    from test_engine_gpu import build_engine
    from maxstyle_amd import engine as E
    e32, W, img, lab, styles = build_engine(dev, spec, B, size, layers)
    e16 = E.InnerLoopEngine(E.NetSpec(spec.reduce, spec.image_ch, spec.num_classes), B, size, size, dev, lr=0.1, act_dtype=BF)
    e16.set_nets(e32.nets)
    slots = {i: E.StyleSlot(i, B, spec.channel_num[i]) for i in layers}
    e16.configure_styles(layers, slots)
    for i in layers:
        st = styles[i]
        e16.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    return e32, e16, img, lab


@pytest.mark.parametrize("net,B,size,ktol", [((4, 1, 4), 4, 64, 1e-2), ((1, 3, 2), 4, 64, 3e-2)])
def test_inner_loop_bf16_storage_vs_fp32_storage(dev, net, B, size, ktol):
    """The whole inner loop with every activation stored as bf16 (InnerLoopEngine(act_dtype=torch.bfloat16): convs, sub-pixel convs, block tails, heads,
    MaxStyle layers, all backward passes) against the fp32-storage engine from the same weights and MaxStyle state.  One evaluation: loss within 1e-2 (2e-3 and 6.5e-3 measured);
    K = 3 free-running: losses within 1 % (3 % for the FCN_64 / 2-class network, whose small-batch trajectory amplifies the first step's 0.65 %), image within
    0.15 (range [0,1]) and rms 0.04; graph replay bit-reproducible; buffers really are bf16.
    (The style GRADIENTS of these random, untrained networks are not compared: the fp32 engine is already 1e-2 from fp64 on them - an amplification of
    1e5 over the fp32 rounding - so a 2^-9 storage rounding saturates it (cosine 0.7-0.9 measured, tools/bf16_grad_check.py); the trained-network test below
    is the meaningful end-to-end statement.)"""
    from oracle import maxstyle_oracle as orc
    spec = orc.NetSpec(*net)
    layers = [3, 4, 5]
    e32, e16, img, lab = _engines(dev, spec, B, size, layers)
    z = e32.encode_fwd(img.to(dev))[0].clone()
    labd = lab.to(dev)
    e32.code, e16.code = z, z.to(BF)
    _, l32 = e32.step_grads(labd)
    _, l16 = e16.step_grads(labd)
    assert abs(float(l32) - float(l16)) < 1e-2 * abs(float(l32)), (float(l32), float(l16))
    outs = []
    for eng in (e32, e16, e16):
        for i in layers:
            eng.styles[i].have_std = False
        st = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
        for i in layers:
            eng.set_style_state(i, st[i].perm, st[i].lmda, st[i].gamma_noise, st[i].beta_noise)
        eng.flat_m.zero_(); eng.flat_v.zero_(); eng.flat_g.zero_()
        out = eng.run(z, labd, 3, use_graph=True).float().clone()
        outs.append((out, eng.losses(3).clone()))
        eng.check_errors(sync=True)
    (o32, ls32), (o16, ls16), (o16b, ls16b) = outs
    assert e16._graph is not None
    assert torch.equal(o16, o16b) and torch.equal(ls16, ls16b)
    assert float(((ls32 - ls16) / ls32).abs().max()) < ktol
    assert float((o32 - o16).abs().max()) < 0.15 and float((o32 - o16).pow(2).mean().sqrt()) < 0.04
    assert e16.buf["d.u4.out"].dtype == BF and e16.buf["s.dh"].dtype == BF and e16.buf["d.u4.bn1.coef"].dtype == torch.float32


def test_bf16_loop_on_trained_networks_vs_reference(dev):
    """K = 5 on the networks TRAINED by the reference's own training step (tests/golden/trained_fcn16.npz), loop on bf16 activation storage, against the
    reference's own run (loop_trained.npz): losses within 3 % of the reference's fp32 losses, image within 8 % of the image range of the reference's fp64
    image (the fp32-storage path is at 1e-6), Dice of the stylised image's segmentation within 2e-2, 99 % of the predicted labels equal.
    The bars are set by the bf16 trajectory's own sensitivity, measured: two runs whose fp32 code z_i differs by 1e-7 (direct against Winograd form of the
    encoder's convolutions) end at losses 0.4 % / 1.8 % and images 3.5 % / 4.9 % from the reference - K = 5 Adam steps at lr 0.1 amplify a rounding."""
    import os
    import numpy as np
    import maxstyle_amd as M
    from oracle import maxstyle_oracle as orc
    from parity_util import rel
    from test_round2_gpu import load_trained
    from test_solver_gpu import injector
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(golden_dir, "loop_trained.npz"))
    W = load_trained(golden_dir)
    spec = orc.NetSpec(4, 1, 4)
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True)
    for name, mod in S.model.items():
        mod.load_state_dict(W[name], strict=True)
        mod.train()
    S.loop_act_dtype = BF
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 777)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = injector(styles, dev)
    z_i, z_s = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    out = S.generate_max_style_image(z_i, layers, spec.channel_num, p=1.5, n_iter=5, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    assert out.dtype == torch.float32
    np.testing.assert_allclose(S.last_losses.cpu().numpy(), g["f32.losses"], rtol=3e-2)
    assert rel(out, g["f64.image"]) < 8e-2
    _, zs2 = S.encode_image(out, disable_track_bn_stats=True)
    logits = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs2, disable_track_bn_stats=True)
    dice = orc.dice_per_class(logits.argmax(1).cpu(), lab, 4)
    np.testing.assert_allclose(dice, g["f32.final_dice"], atol=2e-2)
    assert float((logits.argmax(1).cpu().numpy() == g["f32.final_pred"]).mean()) > 0.99


def test_solver_bf16_loop_through_the_drop_in_api(dev):
    """generate_max_style_image with S.loop_act_dtype = torch.bfloat16: fp32 code in, fp32 image out (reference semantics), the loop inside on bf16 storage;
    module forwards of the same solver (encode_image) stay fp32; result close to the fp32 loop of the same solver."""
    from oracle import maxstyle_oracle as orc
    from test_solver_gpu import make_solver
    spec = orc.NetSpec(4, 1, 4)
    S, W = make_solver(dev, spec)
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 5)
    img, lab = img.to(dev), lab.to(dev)
    z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
    kw = dict(p=1.5, n_iter=3, lr=0.1, reference_image=img, reference_segmentation=lab, fix_seed=11)
    o32 = S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, **kw)
    l32 = S.last_losses.clone()
    S.loop_act_dtype = torch.bfloat16
    o16 = S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, **kw)
    l16 = S.last_losses.clone()
    assert o16.dtype == torch.float32 and o16.shape == o32.shape
    assert float(((l32 - l16) / l32).abs().max()) < 1e-2
    assert float((o32 - o16).abs().max()) < 0.1
    z2, _ = S.encode_image(img, disable_track_bn_stats=True)                 # the module path is untouched by the loop's storage type
    assert z2.dtype == torch.float32 and torch.equal(z2, z_i)
    S.loop_act_dtype = None
    o32b = S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, **kw)
    assert torch.equal(o32b, o32)
