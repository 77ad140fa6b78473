"""bf16 MATRIX arithmetic (`ms_conv2d_bf16m`, `ms_conv2d_actbwd_bf16m`; engine `mfma_bf16=True`): the 3x3 stride-1 convolutions on v_mfma_f32_16x16x16_bf16.

Tolerance statement.  The contraction operands - the prologue's OUTPUT (not its input) and the weights - are rounded to bf16 (relative 2^-9 each) on their way
into LDS; products and sums are fp32.  So against fp64 math on the SAME rounded operands the only error left is the fp32 accumulation and the bf16 rounding of the
stored output: the op-level tests below hold the `_bf16` tests' tolerance (2^-8 relative + 2^-13 of the range) with a margin for prologue values that round
differently in fp32 and fp64 (one bf16 ulp of one operand).  Against UNROUNDED fp32 weights the error is the weights' rounding, ~2^-9 * sqrt(Cin*9) of the terms'
scale - measured at loop level on the trained fixture below."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
    return t.to(BF).to(torch.float32)


def close(got_bf16, ref64, extra):
    g = got_bf16.float().cpu().double()
    err = (g - ref64).abs()
    tol = ref64.abs() * 2.0 ** -8 + ref64.abs().max() * (2.0 ** -13 + extra)
    bad = err > tol
    assert not bool(bad.any()), (int(bad.sum()), float(err.max()), float(ref64.abs().max()))


SHAPES = [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64), (1, 8, 33, 8, 100), (2, 1, 16, 32, 64),      # 64-pixel tiles: NT 1 / 2, tails
          (2, 128, 128, 16, 16), (2, 64, 64, 40, 40), (1, 20, 24, 9, 36), (2, 256, 64, 20, 20),                         # narrow rows (>= 16 pixels) take the kernel too
          (2, 24, 40, 6, 12), (2, 16, 16, 8, 8)]                                                                        # below 16 pixels: the `_bf16` path, fp32 arithmetic


@pytest.mark.parametrize("N,Cin,Cout,H,W", SHAPES)
def test_conv2d_bf16m_all_prologues(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    x = rb(_rand((N, Cin, H, W), 1)); x2 = rb(_rand((N, Cin, H, W), 2)); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4)
    cf = _rand((Cin, 4), 5); cfd = cf.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    bfm = W >= 16 and W % 4 == 0                              # what ms::conv_wide_eligible admits in this mode; below it the `_bf16` kernels run (fp32 arithmetic)
    wb = (rb(w) if bfm else w).double()                       # the kernel rounds the weights when it stages them
    rp = rb if bfm else (lambda t: t)                         # ... and the prologue's output
    a, bb, cc = (cf[:, i].double().view(1, -1, 1, 1) for i in range(3))
    xd, x2d = x.to(dev).to(BF), x2.to(dev).to(BF)
    # plain + bias + BatchNorm statistics epilogue (statistics come from the fp32 accumulators)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, stats=stats, mfma_bf16=True)
    assert out.dtype == BF
    ref = F.conv2d(x.double(), wb, b.double(), padding=1)
    close(out, ref, 0.0)
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
    mean = ref.mean((0, 2, 3)); invstd = 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)
    assert float((coef[:, 2] - mean).abs().max()) < 1e-5 * max(1.0, float(mean.abs().max()))
    assert float((coef[:, 3] / invstd - 1).abs().max()) < 1e-5
    # BatchNorm apply + LeakyReLU prologue: its OUTPUT is what is rounded
    o1 = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2, mfma_bf16=True)
    close(o1, F.conv2d(rp(F.leaky_relu(a * x.double() + bb, 0.2).float()).double(), wb, None, padding=1), 2.0 ** -11)
    # two-tensor BatchNorm-backward prologue + accumulate epilogue
    base = rb(_rand((N, Cout, H, W), 6))
    o2 = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                    pro_cstride=4, in2=x2d, epi_mode=1, out=base.to(dev).to(BF).clone(), mfma_bf16=True)
    close(o2, F.conv2d(rp((a * x.double() + bb * x2.double() + cc).float()).double(), wb, None, padding=1) + base.double(), 2.0 ** -11)
    # same call twice: bit-identical (no atomics in the data path)
    o2b = ops.conv2d(xd, wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                     pro_cstride=4, in2=x2d, epi_mode=1, out=base.to(dev).to(BF).clone(), mfma_bf16=True)
    assert torch.equal(o2, o2b)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 128, 128, 16, 16), (2, 64, 64, 40, 40)])
def test_conv2d_actbwd_bf16m(dev, N, Cin, Cout, H, W):
    """Activation-backward epilogue after the bf16 contraction: masked gradient and the BatchNorm-backward table (fp32 sums of the fp32 masked values)."""
    from maxstyle_amd import ops
    g = rb(_rand((N, Cin, H, W), 21)); w = _rand((Cout, Cin, 3, 3), 23, 0.1)
    u = rb(_rand((N, Cout, H, W), 24) + 0.3)
    coef = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1)
    wp = ops.pack_conv_weight(w.to(dev))
    out, tab = ops.conv2d_actbwd(g.to(dev).to(BF), wp, Cout, 3, u.to(dev).to(BF), coef.to(dev), 0.2, mfma_bf16=True)
    cc = coef.double()
    ref = F.conv2d(g.double(), rb(w).double(), None, padding=1)
    pre = cc[:, 0].view(1, -1, 1, 1) * u.double() + cc[:, 1].view(1, -1, 1, 1)
    safe = (pre.abs() > 1e-4).double()
    refm = ref * torch.where(pre > 0, 1.0, 0.2)
    close((out.float() * safe.float().to(dev)).to(BF), refm * safe, 0.0)
    bc = ops.bn_bwd_coefs(tab, 0, coef.to(dev), N * H * W).cpu().double()
    s1 = refm.sum((0, 2, 3)); s2 = (refm * (u.double() - cc[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
    cnt = N * H * W
    be = -cc[:, 0] * (s2 * cc[:, 3] / cnt) * cc[:, 3]
    ref_bc = torch.stack([cc[:, 0], be, -cc[:, 0] * s1 / cnt - be * cc[:, 2]], 1)
    assert float((bc[:, :3] - ref_bc).abs().max()) < 2e-4 * float(ref_bc.abs().max())


def test_bf16m_differs_from_bf16_only_by_operand_rounding(dev):
    """Against the fp32-arithmetic `_bf16` path with UNROUNDED weights: the gap is the weights' bf16 rounding, a random-sign sum of 2^-9-relative terms."""
    from maxstyle_amd import ops
    N, Cin, Cout, H, W = 2, 64, 64, 64, 64
    x = rb(_rand((N, Cin, H, W), 1)); w = _rand((Cout, Cin, 3, 3), 3, 0.1)
    wp = ops.pack_conv_weight(w.to(dev))
    o_f = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, 3, 1).float()
    o_m = ops.conv2d(x.to(dev).to(BF), wp, None, Cout, 3, 1, mfma_bf16=True).float()
    assert not torch.equal(o_f, o_m)                                           # the bf16 path really ran
    rms = float(o_f.pow(2).mean().sqrt())
    assert float((o_f - o_m).pow(2).mean().sqrt()) < 2.0 ** -7 * rms          # (rounded output: 2^-9 rms each, operand rounding: 2^-9 * O(1))


def test_loop_bf16m_on_trained_networks_vs_reference(dev):
    """K = 5 on the networks trained by the reference's own training step, loop on bf16 storage + bf16 matrix arithmetic, against the reference's own run
    (tests/golden/loop_trained.npz): losses within 3 % of the reference's fp32 losses (measured 1.1 %), image within 8 % of the range of the reference's
    fp64 image (3.3 %), Dice of the stylised image's segmentation within 2e-2 (5e-3), 99 % of the predicted labels equal (99.8 %) - the same bars as bf16
    storage alone (test_bf16_conv_gpu.py: run-to-run sensitivity of a bf16 trajectory 0.4-1.8 % / 3.5-4.9 %): the operand rounding adds nothing visible."""
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
    S.loop_mfma_bf16 = True
    img, lab = orc.synthetic_batch(4, 64, 1, 4, 777)
    layers = [3, 4, 5]
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = injector(styles, dev)
    z_i, z_s = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    out = S.generate_max_style_image(z_i, layers, spec.channel_num, p=1.5, n_iter=5, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    eng = next(iter(S._engines.values()))
    assert eng.mfma_bf16 and eng.bf16 and out.dtype == torch.float32
    losses = S.last_losses.cpu().numpy()
    _, zs2 = S.encode_image(out, disable_track_bn_stats=True)
    logits = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs2, disable_track_bn_stats=True)
    dice = orc.dice_per_class(logits.argmax(1).cpu(), lab, 4)
    agree = float((logits.argmax(1).cpu().numpy() == g["f32.final_pred"]).mean())
    print("bf16m trained fixture: losses", losses, "ref", g["f32.losses"], "image rel", rel(out, g["f64.image"]), "dice", dice, "ref", g["f32.final_dice"], "agree", agree)
    np.testing.assert_allclose(losses, g["f32.losses"], rtol=3e-2)
    assert rel(out, g["f64.image"]) < 8e-2
    np.testing.assert_allclose(dice, g["f32.final_dice"], atol=2e-2)
    assert agree > 0.99


def test_loop_bf16m_graph_replay_is_deterministic_and_close_to_fp32(dev):
    """Random-init FCN_16 at 64x64, K = 3, captured graph: two replays bit-identical; losses within 2 % of the fp32 loop's."""
    from oracle import maxstyle_oracle as orc
    from maxstyle_amd import engine as E
    from test_engine_gpu import build_engine
    spec = orc.NetSpec(4, 1, 4)
    B, size, layers = 4, 64, [3, 4, 5]
    e32, W, img, lab, styles = build_engine(dev, spec, B, size, layers)
    e16 = E.InnerLoopEngine(E.NetSpec(spec.reduce, spec.image_ch, spec.num_classes), B, size, size, dev, lr=0.1, act_dtype=BF, mfma_bf16=True)
    e16.set_nets(e32.nets)
    e16.configure_styles(layers, {i: E.StyleSlot(i, B, spec.channel_num[i]) for i in layers})
    assert e16.mfma_bf16 and not e32.mfma_bf16
    z = e32.encode_fwd(img.to(dev))[0].clone()
    labd = lab.to(dev)
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
    (o32, l32), (o16, l16), (o16b, l16b) = outs
    assert torch.equal(o16, o16b) and torch.equal(l16, l16b)
    assert float(((l32 - l16) / l32).abs().max()) < 2e-2, (l32, l16)
    assert float((o32 - o16).pow(2).mean().sqrt()) < 0.06


def test_mfma_bf16_without_bf16_storage_is_ignored(dev):
    from oracle import maxstyle_oracle as orc
    from maxstyle_amd import engine as E
    e = E.InnerLoopEngine(E.NetSpec(4, 1, 4), 2, 32, 32, dev, mfma_bf16=True)
    assert not e.bf16 and not e.mfma_bf16
