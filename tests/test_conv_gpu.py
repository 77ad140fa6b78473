"""Op-level parity of the HIP conv stack against fp64 PyTorch-CPU math (the same primitives the oracle is made of)."""
import pytest
import torch
import torch.nn.functional as F

from parity_util import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride", [
    (2, 16, 16, 32, 32, 1), (2, 1, 16, 40, 36, 1), (1, 3, 64, 17, 23, 1), (2, 20, 24, 9, 70, 1), (2, 128, 128, 16, 16, 1),
    (3, 32, 16, 64, 64, 1), (2, 16, 16, 32, 32, 2), (1, 64, 64, 24, 40, 2), (2, 128, 128, 32, 32, 2), (2, 256, 96, 8, 8, 1),
])
def test_conv3x3_forward(dev, N, Cin, Cout, H, W, stride):
    from maxstyle_amd import ops
    x = _rand((N, Cin, H, W), 1); w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    out = ops.conv2d(x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), Cout, 3, stride)
    assert out.shape == ref.shape
    assert rel(out, ref) < 2e-6


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 32, 32, 32), (2, 128, 64, 16, 16), (1, 5, 7, 9, 11), (2, 32, 16, 128, 128)])
def test_conv1x1_forward(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    x = _rand((N, Cin, H, W), 1); w = _rand((Cout, Cin, 1, 1), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(x.double(), w.double(), b.double())
    out = ops.conv2d(x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), Cout, 1, 1)
    assert rel(out, ref) < 2e-6


@pytest.mark.parametrize("N,C,Cout,H,W", [(2, 16, 16, 16, 16), (2, 128, 128, 8, 8), (1, 32, 32, 20, 12), (1, 24, 8, 6, 34)])
def test_conv_transpose_2x2(dev, N, C, Cout, H, W):
    from maxstyle_amd import ops
    x = _rand((N, C, H, W), 1); w = _rand((C, Cout, 2, 2), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    out = ops.conv2d(x.to(dev), ops.pack_convT_weight(w.to(dev)), b.to(dev), Cout, 1, 1, epi_mode=2)
    assert out.shape == ref.shape
    assert rel(out, ref) < 2e-6
    # its data-gradient: conv k2 s2
    dy = _rand(ref.shape, 4)
    xr = x.double().requires_grad_(True)
    F.conv_transpose2d(xr, w.double(), b.double(), stride=2).backward(dy.double())
    dx = ops.conv2d(dy.to(dev), ops.pack_convT_weight_dgrad(w.to(dev)), None, C, 2, 2)
    assert rel(dx, xr.grad) < 2e-6


def test_upsample_fused_fetch(dev):
    from maxstyle_amd import ops
    x = _rand((2, 32, 16, 16), 1); w = _rand((16, 32, 3, 3), 2, 0.1); b = _rand((16,), 3)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    out = ops.conv2d(x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), 16, 3, 1, fetch=ops.FETCH_UPS2)
    assert rel(out, ref) < 2e-6


@pytest.mark.parametrize("stride,Cin,Cout,H", [(1, 16, 32, 32), (2, 16, 16, 32), (2, 64, 64, 16), (1, 128, 64, 16)])
def test_conv_data_gradient(dev, stride, Cin, Cout, H):
    """dgrad = the forward kernel on (zero-inserted) dY with flipped/swapped weights, checked against autograd."""
    from maxstyle_amd import ops
    x = _rand((2, Cin, H, H), 1); w = _rand((Cout, Cin, 3, 3), 2, 0.1)
    xr = x.double().requires_grad_(True)
    y = F.conv2d(xr, w.double(), None, stride=stride, padding=1)
    dy = _rand(y.shape, 5)
    y.backward(dy.double())
    fetch = ops.FETCH_NORMAL if stride == 1 else ops.FETCH_ZINS2
    dx = ops.conv2d(dy.to(dev), ops.pack_conv_weight_dgrad(w.to(dev)), None, Cin, 3, 1, fetch=fetch)
    assert dx.shape == x.shape
    assert rel(dx, xr.grad) < 2e-6
    # accumulate epilogue
    base = _rand(x.shape, 6).to(dev)
    acc = ops.conv2d(dy.to(dev), ops.pack_conv_weight_dgrad(w.to(dev)), None, Cin, 3, 1, fetch=fetch, epi_mode=1, out=base.clone())
    assert rel(acc, xr.grad + base.cpu().double()) < 2e-6


@pytest.mark.parametrize("N,C,H,W", [(4, 16, 64, 64), (2, 32, 20, 36), (16, 128, 16, 16)])
def test_batch_stats_and_prologue(dev, N, C, H, W):
    """conv + stats epilogue -> bn_finalize -> next conv with the BN-apply+LeakyReLU prologue == conv(lrelu(bn(conv(x))))."""
    from maxstyle_amd import ops
    x = _rand((N, 8, H, W), 1); w1 = _rand((C, 8, 3, 3), 2, 0.2); b1 = _rand((C,), 3) + 2.0
    gamma = 1 + 0.1 * _rand((C,), 4); beta = 0.1 * _rand((C,), 5)
    w2 = _rand((16, C, 3, 3), 6, 0.1); b2 = _rand((16,), 7)
    u = F.conv2d(x.double(), w1.double(), b1.double(), padding=1)
    mean = u.mean((0, 2, 3)); var = u.var((0, 2, 3), unbiased=False)
    a = F.leaky_relu((u - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1), 0.2)
    ref = F.conv2d(a, w2.double(), b2.double(), padding=1)
    stats, parts = ops.conv_stats_buffer(N, C, H, W, dev)
    ug = ops.conv2d(x.to(dev), ops.pack_conv_weight(w1.to(dev)), b1.to(dev), C, 3, 1, stats=stats)
    coef = ops.bn_finalize(stats, parts, gamma.to(dev), beta.to(dev))
    assert rel(coef[:, 2], mean) < 1e-6
    assert rel(coef[:, 3], 1 / torch.sqrt(var + 1e-5)) < 1e-5
    out = ops.conv2d(ug, ops.pack_conv_weight(w2.to(dev)), b2.to(dev), 16, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(coef)[0], pro_b=ops.coef_ptrs(coef)[1], pro_cstride=4, slope=0.2)
    assert rel(out, ref) < 1e-5
    # materialised BN+residual+activation (same-res and half-res residual)
    res = _rand((N, C, H, W), 8); res_h = _rand((N, C, H // 2, W // 2), 9)
    z = (u - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    o1 = ops.bn_act(ug, coef, res.to(dev), 1, 0.2)
    assert rel(o1, F.leaky_relu(z + res.double(), 0.2)) < 1e-5
    o2 = ops.bn_act(ug, coef, res_h.to(dev), 2, 0.0)
    assert rel(o2, F.relu(z + F.interpolate(res_h.double(), scale_factor=2, mode="nearest"))) < 1e-5


@pytest.mark.parametrize("N,C,H,W", [(4, 16, 32, 32), (2, 16, 16, 128)], ids=["first_generation", "wide_rows"])
def test_bn_backward_chain(dev, N, C, H, W):
    """out = lrelu(s + BN(conv(a))): HIP mask+reduce -> coefs -> dgrad with the BN-backward prologue vs autograd (fp64)."""
    from maxstyle_amd import ops
    a = _rand((N, C, H, W), 1); w = _rand((C, C, 3, 3), 2, 0.15); b = _rand((C,), 3)
    gamma = 1 + 0.1 * _rand((C,), 4); beta = 0.1 * _rand((C,), 5); s = _rand((N, C, H, W), 6); dout = _rand((N, C, H, W), 7)
    ar = a.double().requires_grad_(True)
    u = F.conv2d(ar, w.double(), b.double(), padding=1)
    z = F.batch_norm(u, None, None, gamma.double(), beta.double(), True, 0.0, 1e-5)
    out = F.leaky_relu(s.double() + z, 0.2)
    out.backward(dout.double())
    stats, parts = ops.conv_stats_buffer(N, C, H, W, dev)
    ug = ops.conv2d(a.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), C, 3, 1, stats=stats)
    coef = ops.bn_finalize(stats, parts, gamma.to(dev), beta.to(dev))
    og = ops.bn_act(ug, coef, s.to(dev), 1, 0.2)
    assert rel(og, out) < 1e-5
    g, part, nparts = ops.act_bwd_reduce(dout.to(dev), og, ug, coef, 0.2)
    bc = ops.bn_bwd_coefs(part, nparts, coef, N * H * W)
    da = ops.conv2d(g, ops.pack_conv_weight_dgrad(w.to(dev)), None, C, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(bc)[0], pro_b=ops.coef_ptrs(bc)[1],
                    pro_c=ops.coef_ptrs(bc)[2], pro_cstride=4, in2=ug)
    assert rel(da, ar.grad) < 2e-5
    # un-materialised activation: mask from coef*u+shift
    z2 = F.leaky_relu(F.batch_norm(u.detach(), None, None, gamma.double(), beta.double(), True, 0.0, 1e-5), 0.2)
    g2, _, _ = ops.act_bwd_reduce(dout.to(dev), None, ug, coef, 0.2)
    mask = torch.where(z2 > 0, 1.0, 0.2)
    assert rel(g2, dout.double() * mask) < 1e-6


def test_pool2_sum(dev):
    from maxstyle_amd import ops
    x = _rand((2, 5, 12, 20), 1)
    xr = x.double()
    ref = xr.view(2, 5, 6, 2, 10, 2).sum((3, 5))
    assert rel(ops.pool2_sum(x.to(dev)), ref) < 1e-6
    base = _rand((2, 5, 6, 10), 2).to(dev)
    assert rel(ops.pool2_sum(x.to(dev), out=base.clone(), accumulate=True), ref + base.cpu().double()) < 1e-6


def test_heads(dev):
    from maxstyle_amd import ops
    from oracle import maxstyle_oracle as orc
    N, C, H, W = 3, 16, 24, 20
    h = _rand((N, C, H, W), 1)
    # image head: 1x1 conv + sigmoid and its backward
    w = _rand((1, C), 2, 0.3); b = _rand((1,), 3)
    hr = h.double().requires_grad_(True)
    o = torch.sigmoid(F.conv2d(hr, w.double().view(1, C, 1, 1), b.double()))
    do = _rand(o.shape, 4)
    o.backward(do.double())
    og = ops.head_fwd(h.to(dev), w.to(dev), b.to(dev), True)
    assert rel(og, o) < 1e-6
    assert rel(ops.head_bwd(do.to(dev), og, w.to(dev), C, True), hr.grad) < 1e-5
    # segmentation head + (negated) cross entropy, fused backward
    K = 4
    w4 = _rand((K, C), 5, 0.3); b4 = _rand((K,), 6)
    lab = torch.randint(0, K, (N, H, W), generator=torch.Generator().manual_seed(7))
    hr = h.double().requires_grad_(True)
    logits = F.conv2d(hr, w4.double().view(K, C, 1, 1), b4.double())
    loss = -orc.cross_entropy_2d(logits, lab)
    loss.backward()
    lo, dh, lg = ops.head_ce(h.to(dev), w4.to(dev), b4.to(dev), lab.to(dev), loss_sign=-1.0, need_logits=True)
    assert abs(float(lo) - float(loss)) < 1e-6 * abs(float(loss))
    assert rel(lg, logits) < 1e-6
    assert rel(dh, hr.grad) < 1e-5


@pytest.mark.parametrize("N,C,K,H,W", [(2, 64, 2, 40, 40), (1, 16, 4, 7, 9), (2, 16, 3, 256, 64), (1, 64, 2, 320, 320)])
def test_head_ce_shapes(dev, N, C, K, H, W):
    """ms_head_ce against fp64 math beyond test_heads' one shape: config 4's 64-channel / 2-class head (also grid-stride over a 320 x 320 plane), three classes, a ragged
    7 x 9 plane."""
    from maxstyle_amd import ops
    from oracle import maxstyle_oracle as orc
    h = _rand((N, C, H, W), 11)
    w = _rand((K, C), 12, 0.3); b = _rand((K,), 13)
    lab = torch.randint(0, K, (N, H, W), generator=torch.Generator().manual_seed(14))
    hr = h.double().requires_grad_(True)
    logits = F.conv2d(hr, w.double().view(K, C, 1, 1), b.double())
    loss = -orc.cross_entropy_2d(logits, lab)
    loss.backward()
    lo, dh, lg = ops.head_ce(h.to(dev), w.to(dev), b.to(dev), lab.to(dev), loss_sign=-1.0, need_logits=True)
    assert abs(float(lo) - float(loss)) < 2e-6 * abs(float(loss))
    assert rel(lg, logits) < 2e-6
    assert rel(dh, hr.grad) < 1e-5


WIDE_CASES = [
    # N, Cin, Cout, H, W: rows >= 64 pixels wide take the wide-read kernel (ms_conv_wide.h)
    (2, 16, 16, 64, 64), (1, 16, 16, 9, 128), (2, 1, 16, 30, 72), (2, 20, 24, 13, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64), (1, 8, 33, 7, 100),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", WIDE_CASES)
def test_wide_kernel_all_modes(dev, N, Cin, Cout, H, W):
    """The wide-read 3x3 kernel: plain forward + BatchNorm statistics, BN-apply prologue, BN-backward two-tensor prologue, accumulate
    epilogue - each against fp64 math, and against the first-generation kernel (library option "conv.wide" = 0 is the A/B switch for timing only)."""
    from maxstyle_amd import ops
    x = _rand((N, Cin, H, W), 11); w = _rand((Cout, Cin, 3, 3), 12, 0.1); b = _rand((Cout,), 13)
    wp = ops.pack_conv_weight(w.to(dev))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(x.to(dev), wp, b.to(dev), Cout, 3, 1, stats=stats)
    assert rel(out, ref) < 2e-6
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev))
    assert rel(coef[:, 2], ref.mean((0, 2, 3))) < 1e-5
    assert rel(coef[:, 3], 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-5
    # prologue 1: LeakyReLU(a*x + b) per input channel; zero padding pads the ACTIVATED tensor (b != 0 would leak otherwise)
    cf = _rand((Cin, 4), 14); cf[:, 1] += 0.5
    xa = F.leaky_relu(cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1), 0.2)
    cfd = cf.to(dev)
    o1 = ops.conv2d(x.to(dev), wp, b.to(dev), Cout, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
    assert rel(o1, F.conv2d(xa, w.double(), b.double(), padding=1)) < 3e-6
    # prologue 2: a*x + b*x2 + c (BatchNorm backward), no bias, accumulate into an existing tensor
    x2 = _rand((N, Cin, H, W), 15)
    xb = cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1) * x2.double() + cf[:, 2].double().view(1, -1, 1, 1)
    base = _rand((N, Cout, H, W), 16)
    o2 = ops.conv2d(x.to(dev), wp, None, Cout, 3, 1, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                    pro_cstride=4, in2=x2.to(dev), epi_mode=1, out=base.to(dev).clone())
    assert rel(o2, F.conv2d(xb, w.double(), None, padding=1) + base.double()) < 3e-6


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [
    (2, 16, 16, 64, 64, 3), (1, 32, 48, 20, 192, 3), (2, 64, 64, 64, 64, 3), (1, 8, 33, 7, 100, 3),          # wide-read kernel, NT 1 and 2, channel tails
    (2, 16, 16, 32, 32, 3), (2, 128, 128, 16, 16, 3), (4, 128, 128, 8, 8, 3), (1, 20, 24, 9, 70, 3),          # first-generation kernel: wide and narrow tiles
    (2, 24, 40, 6, 10, 3), (1, 5, 7, 3, 5, 3), (2, 32, 16, 12, 20, 1),                                        # ragged rows (scalar epilogue), 1x1
])
def test_conv_actbwd_epilogue(dev, N, Cin, Cout, H, W, ks):
    """ms_conv2d_actbwd == ms_conv2d followed by ms_act_bwd_reduce(ref = NULL): same masked gradient bit for bit, same BatchNorm-backward
    coefficients up to summation order; with the BatchNorm-backward prologue the data-gradient convs use, and against fp64 math."""
    from maxstyle_amd import ops
    g = _rand((N, Cin, H, W), 21); g2 = _rand((N, Cin, H, W), 22); w = _rand((Cout, Cin, ks, ks), 23, 0.1)
    u = _rand((N, Cout, H, W), 24) + 0.3
    cf = _rand((Cin, 4), 25).to(dev)
    coef = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1).to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    for pro in (0, 2):
        kw = dict(pro_mode=2, pro_a=ops.coef_ptrs(cf)[0], pro_b=ops.coef_ptrs(cf)[1], pro_c=ops.coef_ptrs(cf)[2], pro_cstride=4, in2=g2.to(dev)) if pro else {}
        da = ops.conv2d(g.to(dev), wp, None, Cout, ks, 1, **kw)
        gm, part, nparts = ops.act_bwd_reduce(da, None, u.to(dev), coef, 0.2)
        bc = ops.bn_bwd_coefs(part, nparts, coef, N * H * W)
        out, tab = ops.conv2d_actbwd(g.to(dev), wp, Cout, ks, u.to(dev), coef, 0.2, **kw)
        assert torch.equal(out, gm)
        bc_f = ops.bn_bwd_coefs(tab, 0, coef, N * H * W)
        assert torch.isfinite(bc_f).all()
        tol = 1e-5 * float(gm.abs().mean()) * max(1.0, float(coef[:, 0].abs().max()))
        assert float((bc_f - bc).abs().max()) < 20 * tol + 1e-6 * float(bc.abs().max())
        # fp64 reference of the whole op
        xin = g.double() if not pro else (cf[:, 0].cpu().double().view(1, -1, 1, 1) * g.double() + cf[:, 1].cpu().double().view(1, -1, 1, 1) * g2.double()
                                          + cf[:, 2].cpu().double().view(1, -1, 1, 1))
        ref = F.conv2d(xin, w.double(), None, padding=ks // 2)
        cc = coef.cpu().double()
        pre = cc[:, 0].view(1, -1, 1, 1) * u.double() + cc[:, 1].view(1, -1, 1, 1)
        safe = pre.abs() > 1e-4                                  # away from the activation's kink the mask cannot differ
        refm = ref * torch.where(pre > 0, 1.0, 0.2)
        assert rel(out.cpu().double() * safe, refm * safe) < 3e-6
        s1 = refm.sum((0, 2, 3)); s2 = (refm * (u.double() - cc[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
        cnt = N * H * W
        be = -cc[:, 0] * (s2 * cc[:, 3] / cnt) * cc[:, 3]
        ref_bc = torch.stack([cc[:, 0], be, -cc[:, 0] * s1 / cnt - be * cc[:, 2]], 1)
        assert rel(bc_f[:, :3], ref_bc) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,H,W,up2", [(2, 16, 16, 32, 32, 0), (2, 16, 32, 64, 64, 0), (1, 5, 7, 9, 11, 0), (2, 128, 64, 16, 16, 0), (3, 64, 32, 8, 24, 0),
                                                (2, 16, 16, 32, 32, 1), (2, 128, 64, 8, 8, 1), (1, 20, 24, 9, 14, 1), (1, 6, 5, 7, 9, 1)])
def test_conv1x1_bnres_block_tail(dev, N, Cin, Cout, H, W, up2):
    """ms_conv1x1_bnres: the residual block's tail `lrelu(bn(u) + conv_input(x))` (encoder_decoder.py:62-64, 344-346) as one launch - against fp64 math
    and, bit for bit, against the two launches it replaces (ms_conv2d ks=1 -> ms_bn_act)."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    x = _rand((N, Cin, H, W), 1); w = _rand((Cout, Cin, 1, 1), 2, 0.2); b = _rand((Cout,), 3)
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    u = _rand((N, Cout, Ho, Wo), 4)
    coef = torch.stack([_rand((Cout,), 5).abs() + 0.5, _rand((Cout,), 6), torch.zeros(Cout), torch.ones(Cout)], dim=1).contiguous()
    s = F.conv2d(x.double(), w.double(), b.double())
    if up2:
        s = F.interpolate(s, scale_factor=2, mode="nearest")
    ref = F.leaky_relu(coef[:, 0].double().view(1, -1, 1, 1) * u.double() + coef[:, 1].double().view(1, -1, 1, 1) + s, 0.2)
    xd, ud, cd, bd = x.to(dev), u.to(dev), coef.to(dev), b.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    out = torch.empty(N, Cout, Ho, Wo, device=dev)
    check(lib.ms_conv1x1_bnres(xd.data_ptr(), out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, ud.data_ptr(), cd.data_ptr(), 0.2, up2,
                               torch.cuda.current_stream().cuda_stream), "ms_conv1x1_bnres")
    assert rel(out, ref) < 3e-6
    sd = ops.conv2d(xd, wp, bd, Cout, 1, 1)
    two = torch.empty_like(out)
    check(lib.ms_bn_act(ud.data_ptr(), cd.data_ptr(), sd.data_ptr(), 2 if up2 else 1, two.data_ptr(), N, Cout, Ho, Wo, 0.2, torch.cuda.current_stream().cuda_stream), "ms_bn_act")
    assert torch.equal(out, two), "the fused tail must reproduce conv1x1 + bn_act bit for bit"


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 16, 16, 16), (2, 64, 32, 8, 8), (1, 128, 64, 5, 12), (3, 32, 16, 24, 40), (2, 20, 24, 9, 36), (16, 16, 16, 128, 128)])
def test_subpixel_upsample_conv(dev, N, Cin, Cout, H, W):
    """ms_conv_subpix mode 0 == nn.UpsamplingNearest2d(2) -> Conv2d(3x3, p=1) (encoder_decoder.py:298-300, 323-337) incl. the BatchNorm statistics of the outputs."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    x = _rand((N, Cin, H, W), 1); w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    xd, bd = x.to(dev), b.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    out = torch.empty(N, Cout, 2 * H, 2 * W, device=dev)
    stats, parts = ops.conv_stats_buffer(N, Cout, 2 * H, 2 * W, dev)
    assert lib.ms_conv_subpix_eligible(H, W) == 1
    check(lib.ms_conv_subpix(xd.data_ptr(), out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 0, stats.data_ptr(), 0, 0, 0, 1.0, 0,
                             torch.cuda.current_stream().cuda_stream), "ms_conv_subpix")
    assert rel(out, ref) < 3e-6
    gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    coef = torch.empty(Cout, 4, device=dev)
    check(lib.ms_bn_finalize(stats.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, coef.data_ptr(), Cout, torch.cuda.current_stream().cuda_stream), "fin")
    mean = ref.mean(dim=(0, 2, 3)); var = ref.var(dim=(0, 2, 3), unbiased=False)
    assert rel(coef[:, 2], mean) < 1e-5
    assert rel(coef[:, 3], 1.0 / torch.sqrt(var + 1e-5)) < 1e-5
    # and against the fused-fetch kernel it replaces
    old = ops.conv2d(xd, wp, bd, Cout, 3, 1, fetch=ops.FETCH_UPS2)
    assert rel(out, old) < 2e-6


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 16, 32, 32), (2, 32, 64, 16, 16), (1, 128, 128, 10, 24), (3, 24, 20, 18, 72), (16, 16, 16, 256, 256),
                                            # stored gradients whose width is even but not a multiple of 4 (round 5: the 14-pixel level of the shipped 224-pixel workload;
                                            # the block geometry with a half block at the end of every row)
                                            (20, 128, 128, 28, 28), (3, 24, 20, 18, 36), (1, 16, 16, 4, 4), (2, 40, 8, 6, 12)])
def test_subpixel_stride2_data_gradient(dev, N, Cin, Cout, H, W):
    """ms_conv_subpix mode 1 == d/dx of Conv2d(3x3, s=2, p=1)(x) (res_convdown.down), plain and with the activation-backward epilogue
    (== ms_act_bwd_reduce on the plain result: masked gradient bit for bit, BatchNorm-backward coefficients to rounding)."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    Ho, Wo = H // 2, W // 2
    w = _rand((Cout, Cin, 3, 3), 2, 0.1); g = _rand((N, Cout, Ho, Wo), 5)
    ref = F.conv_transpose2d(g.double(), w.double(), stride=2, padding=1, output_padding=1)
    assert ref.shape == (N, Cin, H, W)
    gd = g.to(dev)
    dwp = ops.pack_conv_weight_dgrad(w.to(dev))
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(N, Cin, H, W, device=dev)
    check(lib.ms_conv_subpix(gd.data_ptr(), out.data_ptr(), dwp.data_ptr(), 0, N, Cout, Ho, Wo, Cin, 1, 0, 0, 0, 0, 1.0, 0, st), "ms_conv_subpix")
    assert rel(out, ref) < 3e-6
    old = ops.conv2d(gd, dwp, None, Cin, 3, 1, fetch=ops.FETCH_ZINS2)
    assert rel(out, old) < 2e-6
    # activation-backward epilogue: mask by the materialised activation `act`, sums against its raw BatchNorm input `u`
    u = _rand((N, Cin, H, W), 7).to(dev)
    coef = torch.stack([_rand((Cin,), 8).abs() + 0.5, _rand((Cin,), 9), _rand((Cin,), 10) * 0.1, torch.rand(Cin) + 0.5], dim=1).contiguous().to(dev)
    act = _rand((N, Cin, H, W), 11).to(dev)
    tab = torch.zeros(lib.ms_conv_actbwd_tab_bytes(Cin) // 4, device=dev)
    masked = torch.empty_like(out)
    check(lib.ms_conv_subpix(gd.data_ptr(), masked.data_ptr(), dwp.data_ptr(), 0, N, Cout, Ho, Wo, Cin, 1, 0, act.data_ptr(), u.data_ptr(), coef.data_ptr(), 0.2,
                             tab.data_ptr(), st), "ms_conv_subpix(actbwd)")
    expect = out * torch.where(act > 0, torch.ones_like(act), torch.full_like(act, 0.2))
    assert torch.equal(masked, expect)
    bc = torch.empty(Cin, 4, device=dev)
    check(lib.ms_bn_bwd_coefs(tab.data_ptr(), 0, coef.data_ptr(), float(N * H * W), bc.data_ptr(), Cin, st), "bn_bwd_coefs")
    nparts = lib.ms_act_bwd_parts(N, Cin, H * W)
    part = torch.empty(Cin, nparts, 2, device=dev)
    g2 = out.clone()
    check(lib.ms_act_bwd_reduce(g2.data_ptr(), act.data_ptr(), u.data_ptr(), coef.data_ptr(), g2.data_ptr(), part.data_ptr(), N, Cin, H * W, 0.2, st), "act_bwd_reduce")
    bc_ref = torch.empty(Cin, 4, device=dev)
    check(lib.ms_bn_bwd_coefs(part.data_ptr(), nparts, coef.data_ptr(), float(N * H * W), bc_ref.data_ptr(), Cin, st), "bn_bwd_coefs")
    assert torch.equal(g2, masked)
    assert rel(bc, bc_ref) < 2e-5
    # ref == NULL: the activation lrelu(sc*u + sh) was never materialised - the mask is recomputed from u (== ms_act_bwd_reduce with ref = NULL)
    masked2 = torch.empty_like(out)
    check(lib.ms_conv_subpix(gd.data_ptr(), masked2.data_ptr(), dwp.data_ptr(), 0, N, Cout, Ho, Wo, Cin, 1, 0, 0, u.data_ptr(), coef.data_ptr(), 0.2,
                             tab.data_ptr(), st), "ms_conv_subpix(actbwd, mask from u)")
    g3 = out.clone()
    check(lib.ms_act_bwd_reduce(g3.data_ptr(), 0, u.data_ptr(), coef.data_ptr(), g3.data_ptr(), part.data_ptr(), N, Cin, H * W, 0.2, st), "act_bwd_reduce(mask from u)")
    assert torch.equal(g3, masked2)


@pytest.mark.parametrize("N,Cin,Cout,Hs,Ws", [(2, 16, 16, 16, 16), (16, 128, 64, 20, 20), (3, 20, 24, 9, 36), (1, 128, 64, 5, 12), (5, 40, 48, 40, 40), (2, 64, 32, 8, 8),
                                              (4, 12, 16, 18, 44), (16, 16, 16, 64, 64)])
@pytest.mark.parametrize("mode", [0, 1])
def test_subpixel_generations_same_bits(dev, N, Cin, Cout, Hs, Ws, mode):
    """Second generation of the sub-pixel kernel (csrc/ms_conv_subpix2.h: LDS-DMA staging, sums appendix, tile and 4 x 4-block work items) against the first: the output
    tensor bit for bit in both geometries (same products in the same order per element), the BatchNorm statistics / activation-backward sums to summation order; and
    against fp64 math.  Shapes: channel tails (Cin % 8 != 0), partial blocks and tiles (Hs % 4 != 0), block groups that cross images, a batch with more items than CUs."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, Hs, Ws), 1).to(dev)
    FIRST, TILES, BLOCKS = 1, 2, 4
    if mode == 0:
        w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
        ref = F.conv2d(F.interpolate(x.cpu().double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
        wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
        sums = torch.empty(int(lib.ms_subpix_pack_floats(Cin, Cout)), device=dev)
        check(lib.ms_subpix_pack(wp.data_ptr(), sums.data_ptr(), Cin, Cout, st), "ms_subpix_pack")
        outs, coefs = {}, {}
        for flags in (FIRST, TILES, BLOCKS, 0):
            out = torch.full((N, Cout, 2 * Hs, 2 * Ws), float("nan"), device=dev)
            stats, parts = ops.conv_stats_buffer(N, Cout, 2 * Hs, 2 * Ws, dev)
            check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), sums.data_ptr(), bd.data_ptr(), N, Cin, Hs, Ws, Cout, 0, stats.data_ptr(), 0, 0, 0, 1.0, 0,
                                      flags, st), "ms_conv_subpix2")
            coef = torch.empty(Cout, 4, device=dev)
            check(lib.ms_bn_finalize(stats.data_ptr(), parts, torch.ones(Cout, device=dev).data_ptr(), torch.zeros(Cout, device=dev).data_ptr(), 1e-5, coef.data_ptr(), Cout, st), "fin")
            outs[flags], coefs[flags] = out, coef
        assert rel(outs[FIRST], ref) < 3e-6
        for flags in (TILES, BLOCKS, 0):
            assert torch.equal(outs[flags], outs[FIRST]), f"flags {flags}: the output must have the first generation's bits"
            assert rel(coefs[flags][:, 2:], coefs[FIRST][:, 2:]) < 1e-5
        # w_sums == NULL: the first generation runs (no appendix to read)
        out = torch.empty_like(outs[FIRST])
        check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), 0, bd.data_ptr(), N, Cin, Hs, Ws, Cout, 0, 0, 0, 0, 0, 1.0, 0, BLOCKS, st), "ms_conv_subpix2(no sums)")
        assert torch.equal(out, outs[FIRST])
        return
    # mode 1: data-gradient of Conv2d(3x3, s=2, p=1) from Cout_fwd = Cin (of this call) gradient channels to Cin_fwd = Cout channels
    w = _rand((Cin, Cout, 3, 3), 2, 0.1)
    ref = F.conv_transpose2d(x.cpu().double(), w.double(), stride=2, padding=1, output_padding=1)
    dwp = ops.pack_conv_weight_dgrad(w.to(dev))
    u = _rand((N, Cout, 2 * Hs, 2 * Ws), 7).to(dev)
    coef = torch.stack([_rand((Cout,), 8).abs() + 0.5, _rand((Cout,), 9), _rand((Cout,), 10) * 0.1, torch.rand(Cout) + 0.5], dim=1).contiguous().to(dev)
    act = _rand((N, Cout, 2 * Hs, 2 * Ws), 11).to(dev)
    res = {}
    for flags in (FIRST, TILES, BLOCKS, 0):
        plain = torch.full((N, Cout, 2 * Hs, 2 * Ws), float("nan"), device=dev)
        check(lib.ms_conv_subpix2(x.data_ptr(), plain.data_ptr(), dwp.data_ptr(), 0, 0, N, Cin, Hs, Ws, Cout, 1, 0, 0, 0, 0, 1.0, 0, flags, st), "ms_conv_subpix2")
        masked = torch.full_like(plain, float("nan"))
        tab = torch.zeros(lib.ms_conv_actbwd_tab_bytes(Cout) // 4, device=dev)
        check(lib.ms_conv_subpix2(x.data_ptr(), masked.data_ptr(), dwp.data_ptr(), 0, 0, N, Cin, Hs, Ws, Cout, 1, 0, act.data_ptr(), u.data_ptr(), coef.data_ptr(), 0.2,
                                  tab.data_ptr(), flags, st), "ms_conv_subpix2(actbwd)")
        bc = torch.empty(Cout, 4, device=dev)
        check(lib.ms_bn_bwd_coefs(tab.data_ptr(), 0, coef.data_ptr(), float(N * 4 * Hs * Ws), bc.data_ptr(), Cout, st), "bn_bwd_coefs")
        res[flags] = (plain, masked, bc)
    assert rel(res[FIRST][0], ref) < 3e-6
    for flags in (TILES, BLOCKS, 0):
        assert torch.equal(res[flags][0], res[FIRST][0]) and torch.equal(res[flags][1], res[FIRST][1]), f"flags {flags}"
        assert rel(res[flags][2], res[FIRST][2]) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(16, 512, 512, 40, 40), (16, 64, 64, 64, 64), (4, 128, 96, 80, 80), (3, 32, 48, 36, 72), (2, 20, 16, 18, 40), (16, 16, 16, 64, 128),
                                            (5, 64, 160, 24, 24), (16, 128, 128, 32, 32)])
def test_stride2_conv_second_generation(dev, N, Cin, Cout, H, W):
    """3x3 stride-2 forward conv (res_convdown.down, encoder_decoder.py:40), second generation (csrc/ms_conv_s2.h) against fp64 math and against the first generation
    (same products; the order of the 4-channel groups can differ where the first generation uses 8-channel chunks: rounding level).  Shapes: both work-item geometries
    (tiles / 4 x 4 blocks: small outputs), 1 / 2 / 4 channel blocks, partial tiles and blocks (Hout % 4 != 0), channel counts that are not multiples of 16."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(x.cpu().double(), w.double(), b.double(), stride=2, padding=1)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    Ho, Wo = ref.shape[2], ref.shape[3]

    def run():
        out = torch.full((N, Cout, Ho, Wo), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 3, 2, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d(s2)")
        return out
    was = lib.ms_set_option(b"conv.s2g2", 1)
    try:
        new = run()
        lib.ms_set_option(b"conv.s2g2", 0)
        old = run()
    finally:
        lib.ms_set_option(b"conv.s2g2", was)
    assert rel(old, ref) < 3e-6
    assert rel(new, ref) < 3e-6
    assert rel(new, old) < 2e-6
    assert torch.equal(new, run())              # and deterministic


@pytest.mark.parametrize("N,Cin,Cout,H,W,pro", [(2, 16, 1, 32, 64, 2), (16, 16, 1, 256, 256, 2), (2, 64, 3, 40, 72, 2), (1, 7, 2, 19, 20, 0), (3, 16, 4, 16, 128, 0)])
def test_small_cout_conv(dev, N, Cin, Cout, H, W, pro):
    """ms_conv3x3_small_cout (vector-ALU 3x3 conv for <= 4 output channels: the data-gradient to the image) vs fp64 math and vs ms_conv2d."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    x = _rand((N, Cin, H, W), 1); u = _rand((N, Cin, H, W), 2); w = _rand((Cout, Cin, 3, 3), 3, 0.2)
    bc = _rand((Cin, 4), 4)
    if pro == 2:
        p = bc[:, 0].double().view(1, -1, 1, 1) * x.double() + bc[:, 1].double().view(1, -1, 1, 1) * u.double() + bc[:, 2].double().view(1, -1, 1, 1)
    else:
        p = x.double()
    ref = F.conv2d(p, w.double(), None, padding=1)
    xd, ud, bcd = x.to(dev), u.to(dev), bc.to(dev).contiguous()
    wp = ops.pack_conv_weight(w.to(dev))
    out = torch.empty(N, Cout, H, W, device=dev)
    pa, pb, pc = ops.coef_ptrs(bcd)
    assert lib.ms_conv3x3_small_cout_ok(Cout, W) == 1
    check(lib.ms_conv3x3_small_cout(xd.data_ptr(), ud.data_ptr() if pro == 2 else 0, out.data_ptr(), wp.data_ptr(), N, Cin, H, W, Cout, pro,
                                    pa if pro == 2 else 0, pb if pro == 2 else 0, pc if pro == 2 else 0, 4, torch.cuda.current_stream().cuda_stream), "small_cout")
    assert rel(out, ref) < 3e-6
    kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=ud) if pro == 2 else {}
    old = ops.conv2d(xd, wp, None, Cout, 3, 1, **kw)
    assert rel(out, old) < 3e-6


def test_wide_kernel_8_row_tiles_in_a_subprocess():
    """The 8-row-tile variant of the wide kernel (two output rows per MFMA wave) is selected by the library option "conv.wide_rows" = 8: run the wide-kernel and
    activation-backward cases in a child process started with that option (the MS_OPTIONS harness hook of maxstyle_amd/options.py)."""
    import os, subprocess, sys
    env = dict(os.environ, MS_OPTIONS="conv.wide_rows=8")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_conv_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "wide_kernel_all_modes or conv_actbwd_epilogue or batch_stats_and_prologue or bn_backward_chain"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_activation_slope_range_is_checked(dev):
    """The activation helper computes max(v, v*slope) - exact for the LeakyReLU / ReLU slopes the reference uses (encoder_decoder.py:646,655) and only
    for slopes in [0, 1]: anything else is refused at the entry points, never computed wrong."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import MaxStyleHipError
    u = _rand((1, 4, 8, 8), 1).to(dev)
    coef = torch.ones(4, 4, device=dev)
    with pytest.raises(MaxStyleHipError):
        ops.bn_act(u, coef, slope=1.5)
    with pytest.raises(MaxStyleHipError):
        ops.bn_act(u, coef, slope=-0.1)
    w = ops.pack_conv_weight(_rand((4, 4, 3, 3), 2, 0.1).to(dev))
    with pytest.raises(MaxStyleHipError):
        ops.conv2d(u, w, None, 4, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(coef)[0], pro_b=ops.coef_ptrs(coef)[1], pro_cstride=4, slope=2.0)
    out = ops.bn_act(u, coef, slope=0.0)                                   # ReLU
    assert torch.equal(out, torch.relu(u + 1.0))
    neg = ops.bn_act(u, coef, slope=0.2)
    assert torch.equal(neg, torch.nn.functional.leaky_relu(u + 1.0, 0.2))   # max(v, 0.2 v) == the select form, bit for bit
