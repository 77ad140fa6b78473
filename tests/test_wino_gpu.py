"""Winograd F(2x2,3x3) form of the wide 3x3 stride-1 convolution (fetch | FETCH_WINOGRAD; ms_conv_wide.h, storage tag ms_f32w).

Tolerance statement: against fp64 on random data the form is as close as the direct one (measured 1-4e-7 of the output range on every shape below; bar 2e-6).
On the networks' activations its rounding error is about twice the direct form's (include/maxstyle_hip.h, ms_conv2d); the loop-level parity tests
(test_engine_gpu.py, test_round2_gpu.py, test_solver_gpu.py) run WITH it and hold their bars unchanged; the training passes take it only on request
(MS_TRAIN_WINOGRAD=1: test_train_gpu.py holds its bars under the switch too, tools/test_switches.sh)."""
import os
import sys

import pytest
import torch

from parity_util import set_engine_default
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def err(o, r):
    return float((o.cpu().double() - r).abs().max() / r.abs().max())


SHAPES = [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 64, 64),      # interior tiles only / several channel blocks / a channel tail in Cout
          (2, 64, 64, 32, 32), (2, 128, 64, 20, 40), (1, 16, 16, 9, 36), (2, 64, 32, 20, 20), (1, 16, 48, 11, 28),      # 8-row x 32-pixel tiles (rows of 20..63 pixels)
          (1, 8, 33, 10, 100), (2, 16, 16, 6, 72), (1, 24, 16, 7, 68),         # ragged width, heights that are not multiples of the 4-row tile (odd: half a 2x2 tile)
          (2, 16, 16, 128, 128)]


@pytest.mark.parametrize("N,Cin,Cout,H,W", SHAPES)
def test_winograd_form_all_prologues_and_epilogues(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    WG = ops.FETCH_WINOGRAD
    x = _rand((N, Cin, H, W), 1); x2 = _rand((N, Cin, H, W), 2); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4)
    cf = _rand((Cin, 4), 5); cfd = cf.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    a, bb, cc = (cf[:, i].double().view(1, -1, 1, 1) for i in range(3))
    xd, x2d = x.to(dev), x2.to(dev)
    TOL = 2e-6
    # plain + bias + BatchNorm statistics
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, fetch=WG, stats=stats)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert err(out, ref) < TOL
    direct = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1)
    assert not torch.equal(out, direct)                                    # the other form really ran ...
    assert float((out - direct).abs().max()) < 4e-6 * float(ref.abs().max())      # ... and agrees with the direct one to fp32 rounding
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
    mean = ref.mean((0, 2, 3)); invstd = 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)
    assert float((coef[:, 2] - mean).abs().max()) < 1e-5 * max(1.0, float(mean.abs().max()))
    assert float((coef[:, 3] / invstd - 1).abs().max()) < 1e-5
    # BatchNorm apply + LeakyReLU prologue
    o1 = ops.conv2d(xd, wp, None, Cout, 3, 1, fetch=WG, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
    assert err(o1, F.conv2d(F.leaky_relu(a * x.double() + bb, 0.2), w.double(), None, padding=1)) < TOL
    # two-tensor BatchNorm-backward prologue + accumulate epilogue
    base = _rand((N, Cout, H, W), 6)
    kw = dict(fetch=WG, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2], pro_cstride=4, in2=x2d, epi_mode=1)
    o2 = ops.conv2d(xd, wp, None, Cout, 3, 1, out=base.to(dev).clone(), **kw)
    assert err(o2, F.conv2d(a * x.double() + bb * x2.double() + cc, w.double(), None, padding=1) + base.double()) < TOL
    assert torch.equal(o2, ops.conv2d(xd, wp, None, Cout, 3, 1, out=base.to(dev).clone(), **kw))      # run to run: the same bits
    # activation-backward epilogue (mask tensor through LDS) + its BatchNorm-backward table
    u = _rand((N, Cout, H, W), 24) + 0.3
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1)
    og, tab = ops.conv2d_actbwd(xd, wp, Cout, 3, u.to(dev), coef4.to(dev), 0.2, fetch=WG)
    c4 = coef4.double()
    pre = c4[:, 0].view(1, -1, 1, 1) * u.double() + c4[:, 1].view(1, -1, 1, 1)
    refm = F.conv2d(x.double(), w.double(), None, padding=1) * torch.where(pre > 0, 1.0, 0.2)
    safe = (pre.abs() > 1e-4).double()
    assert err(og.cpu().double() * safe, refm * safe) < TOL
    bc = ops.bn_bwd_coefs(tab, 0, coef4.to(dev), N * H * W).cpu().double()
    s1 = refm.sum((0, 2, 3)); s2 = (refm * (u.double() - c4[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
    cnt = N * H * W
    be = -c4[:, 0] * (s2 * c4[:, 3] / cnt) * c4[:, 3]
    ref_bc = torch.stack([c4[:, 0], be, -c4[:, 0] * s1 / cnt - be * c4[:, 2]], 1)
    assert float((bc[:, :3] - ref_bc).abs().max()) < 2e-4 * float(ref_bc.abs().max())


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 64, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 32, 32), (2, 128, 64, 20, 40), (1, 16, 48, 11, 28), (1, 8, 33, 10, 100),
                                            (2, 32, 32, 128, 128), (1, 64, 128, 40, 40)])
@pytest.mark.parametrize("bf16", [False, True])
def test_two_channel_blocks_per_staged_tile_same_bits(dev, N, Cin, Cout, H, W, bf16):
    """Round 4: layers with more than 16 output channels stage / transform the input tile once per 32 of them (NT = 2, one workgroup per CU on 256 registers).
    Per output element the accumulation order is the one-block variant's (MS_FETCH_WINO_NT1), so every stored tensor is bit-identical - all prologues and
    epilogues, channel tails (48, 33), ragged tiles, both tile shapes, both storage types; the BatchNorm partial tables group their sums by work item and
    agree to rounding."""
    from maxstyle_amd import ops
    WG, N1 = ops.FETCH_WINOGRAD, ops.FETCH_WINOGRAD | ops.FETCH_WINO_NT1
    dt = torch.bfloat16 if bf16 else torch.float32
    x = _rand((N, Cin, H, W), 1).to(dev).to(dt); x2 = _rand((N, Cin, H, W), 2).to(dev).to(dt); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4).to(dev)
    cfd = _rand((Cin, 4), 5).to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    pa, pb, pc = ops.coef_ptrs(cfd)[:3]
    s2, parts = ops.conv_stats_buffer(N, Cout, H, W, dev); s1, _ = ops.conv_stats_buffer(N, Cout, H, W, dev)
    o2 = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=WG, stats=s2); o1 = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=N1, stats=s1)
    assert torch.equal(o1, o2)
    one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    c2, c1 = ops.bn_finalize(s2, parts, one, zero), ops.bn_finalize(s1, parts, one, zero)
    assert float((c2 - c1).abs().max()) < 2e-6 * float(c1.abs().max())
    kw = dict(pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
    assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=WG, **kw), ops.conv2d(x, wp, None, Cout, 3, 1, fetch=N1, **kw))
    base = _rand((N, Cout, H, W), 6).to(dev).to(dt)
    kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2, epi_mode=1)
    assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=WG, out=base.clone(), **kw), ops.conv2d(x, wp, None, Cout, 3, 1, fetch=N1, out=base.clone(), **kw))
    u = (_rand((N, Cout, H, W), 24) + 0.3).to(dev).to(dt)
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1).to(dev)
    kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2)
    g2, t2 = ops.conv2d_actbwd(x, wp, Cout, 3, u, coef4, 0.2, fetch=WG, **kw); g1, t1 = ops.conv2d_actbwd(x, wp, Cout, 3, u, coef4, 0.2, fetch=N1, **kw)
    assert torch.equal(g1, g2)
    b2, b1 = ops.bn_bwd_coefs(t2, 0, coef4, N * H * W), ops.bn_bwd_coefs(t1, 0, coef4, N * H * W)
    assert float((b2 - b1).abs().max()) < 1e-5 * float(b1.abs().max())
    if H % 2 == 0 and ops.lib.ms_conv2d_pool2_ok(N, Cin, H, W, Cout, 0, int(bf16)):
        assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=WG, epi_mode=ops.EPI_POOL2), ops.conv2d(x, wp, None, Cout, 3, 1, fetch=N1, epi_mode=ops.EPI_POOL2))


BLOCK_SHAPES = [(2, 64, 64, 40, 40), (16, 64, 32, 20, 20), (3, 32, 48, 80, 80), (2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 128, 64, 20, 40), (1, 16, 48, 11, 28),
                (1, 8, 33, 10, 100), (2, 16, 16, 6, 72), (1, 24, 16, 7, 68), (5, 16, 16, 24, 24), (1, 64, 64, 24, 36)]


@pytest.mark.parametrize("N,Cin,Cout,H,W", BLOCK_SHAPES)
@pytest.mark.parametrize("bf16", [False, True])
def test_block_form_same_bits_as_the_tiled_form(dev, N, Cin, Cout, H, W, bf16):
    """Round 4: the BLOCK form of the Winograd kernel (MS_FETCH_WINO_BLOCKS: four independent 8x8-pixel blocks per work item, each staged with its own halo, taken by
    the dispatch where 64x4 / 32x8 tiles would be mostly padding - rows of 80, 40, 20 pixels) against the tiled one-block kernel (MS_FETCH_WINO_NT1): per output element
    the same accumulation order, so every stored tensor is bit-identical - all prologues and epilogues, image sizes that are not multiples of the block (partial blocks,
    odd heights), block lists that do not fill the last group of four, several images per group, channel tails, both storage types; BatchNorm tables agree to rounding."""
    from maxstyle_amd import ops
    BL, N1 = ops.FETCH_WINOGRAD | ops.FETCH_WINO_BLOCKS, ops.FETCH_WINOGRAD | ops.FETCH_WINO_NT1
    assert ops.lib.ms_conv2d_form(N, Cin, H, W, Cout, 0, int(bf16), BL) in (4, 5) and ops.lib.ms_conv2d_form(N, Cin, H, W, Cout, 0, int(bf16), N1) == 2
    dt = torch.bfloat16 if bf16 else torch.float32
    x = _rand((N, Cin, H, W), 1).to(dev).to(dt); x2 = _rand((N, Cin, H, W), 2).to(dev).to(dt); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4).to(dev)
    cfd = _rand((Cin, 4), 5).to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    pa, pb, pc = ops.coef_ptrs(cfd)[:3]
    s2, parts = ops.conv_stats_buffer(N, Cout, H, W, dev); s1, _ = ops.conv_stats_buffer(N, Cout, H, W, dev)
    o2 = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=BL, stats=s2); o1 = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=N1, stats=s1)
    assert torch.equal(o1, o2)
    one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    c2, c1 = ops.bn_finalize(s2, parts, one, zero), ops.bn_finalize(s1, parts, one, zero)
    assert float((c2 - c1).abs().max()) < 2e-6 * float(c1.abs().max())
    kw = dict(pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
    assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=BL, **kw), ops.conv2d(x, wp, None, Cout, 3, 1, fetch=N1, **kw))
    base = _rand((N, Cout, H, W), 6).to(dev).to(dt)
    kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2, epi_mode=1)
    assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=BL, out=base.clone(), **kw), ops.conv2d(x, wp, None, Cout, 3, 1, fetch=N1, out=base.clone(), **kw))
    u = (_rand((N, Cout, H, W), 24) + 0.3).to(dev).to(dt)
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1).to(dev)
    kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2)
    g2, t2 = ops.conv2d_actbwd(x, wp, Cout, 3, u, coef4, 0.2, fetch=BL, **kw); g1, t1 = ops.conv2d_actbwd(x, wp, Cout, 3, u, coef4, 0.2, fetch=N1, **kw)
    assert torch.equal(g1, g2)
    b2, b1 = ops.bn_bwd_coefs(t2, 0, coef4, N * H * W), ops.bn_bwd_coefs(t1, 0, coef4, N * H * W)
    assert float((b2 - b1).abs().max()) < 1e-5 * float(b1.abs().max())
    if H % 2 == 0 and ops.lib.ms_conv2d_pool2_ok(N, Cin, H, W, Cout, 0, int(bf16)):
        assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=BL, epi_mode=ops.EPI_POOL2), ops.conv2d(x, wp, None, Cout, 3, 1, fetch=N1, epi_mode=ops.EPI_POOL2))
    # with the Winograd appendix (LDS-DMA of the transformed weights) the block form stays bit-identical too
    wpu, has = ops.with_wino_appendix(wp.clone(), Cin, Cout)
    if has:
        assert torch.equal(ops.conv2d(x, wpu, b, Cout, 3, 1, fetch=BL | ops.FETCH_WINO_U), o1)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(16, 64, 64, 20, 20), (3, 32, 48, 20, 20), (1, 64, 64, 20, 20), (2, 16, 32, 14, 20), (5, 24, 40, 18, 20), (16, 16, 96, 20, 20), (7, 8, 33, 26, 20),
                                            (20, 32, 64, 24, 24), (3, 16, 48, 12, 24), (1, 64, 32, 24, 24), (20, 32, 64, 28, 28), (2, 24, 40, 10, 28), (5, 16, 33, 28, 28),
                                            (64, 16, 256, 20, 20), (48, 8, 256, 28, 28)])
def test_flat_form_same_bits_as_the_tiled_form(dev, N, Cin, Cout, H, W):
    """Round 6: the FLAT form of the Winograd kernel (ms_f32wf_t<W>: the 2x2-output tiles of a batch of 20 / 24 / 28-pixel-wide images as ONE list, 64 consecutive tiles per work item, the band
    of image rows they touch staged per chunk - of one image or across an image boundary; ms_conv2d_form 7) against the 8-row x 32-pixel tiles it replaces (option
    "conv.wino_flat" = 0): per output element the same accumulation order, so every stored tensor is bit-identical - all three prologues, the plain / accumulate /
    activation-backward epilogues, statistics; tile lists that end inside a group, groups that span two images, a batch of one (no next image), channel tails (an odd number of 16-channel blocks: the
    second block of the last item does not exist), the smallest legal heights (64 tiles per image), launches of three to five rounds of the persistent grid (every workgroup runs several items:
    the cross-item operand prefetch); the BatchNorm tables agree to rounding (another grouping of the per-lane sums)."""
    from maxstyle_amd import ops
    from maxstyle_amd.options import library_option
    U = ops.FETCH_WINOGRAD | ops.FETCH_WINO_U
    x = _rand((N, Cin, H, W), 1).to(dev); x2 = _rand((N, Cin, H, W), 2).to(dev); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4).to(dev)
    cfd = _rand((Cin, 4), 5).to(dev)
    pa, pb, pc = ops.coef_ptrs(cfd)[:3]
    wp, has = ops.with_wino_appendix(ops.pack_conv_weight(w.to(dev)), Cin, Cout)
    assert has
    base = _rand((N, Cout, H, W), 6).to(dev)
    u = (_rand((N, Cout, H, W), 24) + 0.3).to(dev)
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1).to(dev)
    one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)

    def run():
        r = {}
        st, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
        r["fwd"] = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U, stats=st)
        r["fwd.coef"] = ops.bn_finalize(st, parts, one, zero)
        st1, _ = ops.conv_stats_buffer(N, Cout, H, W, dev)
        r["pro1"] = ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U, stats=st1, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
        r["pro1.coef"] = ops.bn_finalize(st1, parts, one, zero)
        kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2)
        r["pro2"] = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, **kw)
        r["acc"] = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, out=base.clone(), epi_mode=1, **kw)
        r["acc0"] = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, out=base.clone(), epi_mode=1)
        g, t = ops.conv2d_actbwd(x, wp, Cout, 3, u, coef4, 0.2, fetch=U, **kw)
        r["actbwd"] = g; r["actbwd.coef"] = ops.bn_bwd_coefs(t, 0, coef4, N * H * W)
        torch.cuda.synchronize()
        return r
    with library_option("conv.wino_nt", 2):
        with library_option("conv.wino_flat", 2):       # wherever legal (the default takes it where it saves a round of the persistent grid)
            assert ops.lib.ms_conv2d_form(N, Cin, H, W, Cout, 0, 0, U) == 7
            flat = run()
        with library_option("conv.wino_flat", 0):
            assert ops.lib.ms_conv2d_form(N, Cin, H, W, Cout, 0, 0, U) == 3
            tiled = run()
    for k in ("fwd", "pro1", "pro2", "acc", "acc0", "actbwd"):
        assert torch.equal(flat[k], tiled[k]), (k, float((flat[k] - tiled[k]).abs().max()))
    for k, tol in (("fwd.coef", 2e-6), ("pro1.coef", 2e-6), ("actbwd.coef", 1e-5)):
        assert float((flat[k] - tiled[k]).abs().max()) < tol * float(tiled[k].abs().max()), k
    # and against fp64 (the form's own error: 1-5e-7 of the output range on random data, as the tiled form's)
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double(), b.double().cpu(), padding=1)
    assert float((flat["fwd"].double().cpu() - ref).abs().max()) < 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 16, 64, 64), (2, 64, 64, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 32, 32), (1, 128, 64, 20, 40), (1, 8, 33, 10, 100), (1, 256, 256, 40, 40)])
def test_winograd_appendix_same_bits(dev, N, Cin, Cout, H, W):
    """MS_FETCH_WINO_U: the transformed weights staged by LDS-DMA from the packed tensor's appendix (ms_wino_pack: the in-kernel expression, evaluated once per weight
    version) against the kernel that transforms the nine taps of every chunk itself - the same bits, for one and two channel blocks per staged tile, every prologue,
    channel tails in Cout, several chunks per item (weights re-staged per chunk) and layers whose two chunks stay resident (16 -> 16)."""
    from maxstyle_amd import ops
    WG = ops.FETCH_WINOGRAD
    x = _rand((N, Cin, H, W), 1).to(dev); x2 = _rand((N, Cin, H, W), 2).to(dev); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4).to(dev)
    cfd = _rand((Cin, 4), 5).to(dev)
    pa, pb, pc = ops.coef_ptrs(cfd)[:3]
    wp0 = ops.pack_conv_weight(w.to(dev))
    wp, has = ops.with_wino_appendix(wp0.clone(), Cin, Cout)
    assert has and torch.equal(wp, wp0)
    for nt in (0, ops.FETCH_WINO_NT1):
        U = WG | nt | ops.FETCH_WINO_U
        assert torch.equal(ops.conv2d(x, wp, b, Cout, 3, 1, fetch=U), ops.conv2d(x, wp0, b, Cout, 3, 1, fetch=WG | nt))
        kw = dict(pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
        assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, **kw), ops.conv2d(x, wp0, None, Cout, 3, 1, fetch=WG | nt, **kw))
        kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2)
        assert torch.equal(ops.conv2d(x, wp, None, Cout, 3, 1, fetch=U, **kw), ops.conv2d(x, wp0, None, Cout, 3, 1, fetch=WG | nt, **kw))
    # a change of the taps is followed by the appendix only through wino_repack
    wp.mul_(2.0)
    stale = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=WG | ops.FETCH_WINO_U)
    ops.wino_repack(wp, Cin, Cout)
    fresh = ops.conv2d(x, wp, None, Cout, 3, 1, fetch=WG | ops.FETCH_WINO_U)
    assert torch.equal(fresh, ops.conv2d(x, wp, None, Cout, 3, 1, fetch=WG)) and not torch.equal(stale, fresh)


def test_winograd_bit_is_ignored_where_the_form_is_not_built(dev):
    """Rows narrower than 20 pixels, Cin % 8 != 0, 1x1: the call runs the direct form - bit-identical with and without the bit."""
    from maxstyle_amd import ops
    for (N, Cin, Cout, H, W, ks, stride) in [(2, 16, 16, 16, 16, 3, 1), (2, 12, 16, 16, 64, 3, 1), (2, 16, 16, 64, 64, 1, 1)]:
        x = _rand((N, Cin, H, W), 1).to(dev); w = _rand((Cout, Cin, ks, ks), 3, 0.1)
        wp = ops.pack_conv_weight(w.to(dev))
        assert torch.equal(ops.conv2d(x, wp, None, Cout, ks, stride, fetch=ops.FETCH_WINOGRAD), ops.conv2d(x, wp, None, Cout, ks, stride))


def test_which_engines_ask_for_the_winograd_form(dev, monkeypatch):
    """Both engines ask for the form by default - the inner loop since round 2, the training passes' forward / data-gradient convs since round 5 (their weight-gradient
    fidelity over 20 seeds equals the direct form's: profiles/r05_train_fidelity.json) - and EngineOptions.winograd / .train_winograd = False switch it off per engine."""
    from maxstyle_amd import engine as E
    from maxstyle_amd import options as O
    from maxstyle_amd.train_engine import TrainEngine
    monkeypatch.delitem(O._engine_defaults, "winograd", raising=False); monkeypatch.delitem(O._engine_defaults, "train_winograd", raising=False)
    assert E.InnerLoopEngine(E.NetSpec(4, 1, 4), 2, 64, 64, dev).winograd
    assert TrainEngine(E.NetSpec(4, 1, 4), 2, 64, 64, dev).winograd
    assert not TrainEngine(E.NetSpec(4, 1, 4), 2, 64, 64, dev, options={"train_winograd": False}).winograd
    assert TrainEngine(E.NetSpec(4, 1, 4), 2, 64, 64, dev, options={"winograd": False}).winograd              # (the loop's switch is not the passes')
    set_engine_default(monkeypatch, "winograd", False)
    assert not E.InnerLoopEngine(E.NetSpec(4, 1, 4), 2, 64, 64, dev).winograd


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 16, 64, 64), (1, 32, 48, 20, 192), (2, 64, 64, 32, 32), (1, 8, 33, 10, 100), (1, 16, 16, 9, 36)])
def test_winograd_form_on_bf16_storage(dev, N, Cin, Cout, H, W):
    """bf16 activation storage under the Winograd form (tags ms_bf16w / ms_bf16w32): fp32 arithmetic on the widened values, so against fp64 on the
    ROUNDED inputs the only error beyond fp32 rounding is the bf16 rounding of the stored output (the `_bf16` tests' bar)."""
    from maxstyle_amd import ops
    BF = torch.bfloat16
    WG = ops.FETCH_WINOGRAD

    def rb(t):
        return t.to(BF).to(torch.float32)

    def close(got, ref64):
        g = got.float().cpu().double()
        bad = (g - ref64).abs() > ref64.abs() * 2.0 ** -8 + ref64.abs().max() * 2.0 ** -13
        assert not bool(bad.any()), (int(bad.sum()), float((g - ref64).abs().max()), float(ref64.abs().max()))
    x = rb(_rand((N, Cin, H, W), 1)); x2 = rb(_rand((N, Cin, H, W), 2)); w = _rand((Cout, Cin, 3, 3), 3, 0.1); b = _rand((Cout,), 4)
    cf = _rand((Cin, 4), 5); cfd = cf.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    a, bb, cc = (cf[:, i].double().view(1, -1, 1, 1) for i in range(3))
    xd, x2d = x.to(dev).to(BF), x2.to(dev).to(BF)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    out = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, fetch=WG, stats=stats)
    assert out.dtype == BF
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    close(out, ref)
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)).cpu().double()
    mean = ref.mean((0, 2, 3)); invstd = 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)
    assert float((coef[:, 2] - mean).abs().max()) < 1e-5 * max(1.0, float(mean.abs().max()))      # statistics of the fp32 values before rounding
    assert float((coef[:, 3] / invstd - 1).abs().max()) < 1e-5
    o1 = ops.conv2d(xd, wp, None, Cout, 3, 1, fetch=WG, pro_mode=1, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_cstride=4, slope=0.2)
    close(o1, F.conv2d(F.leaky_relu(a * x.double() + bb, 0.2), w.double(), None, padding=1))
    base = rb(_rand((N, Cout, H, W), 6))
    o2 = ops.conv2d(xd, wp, None, Cout, 3, 1, fetch=WG, pro_mode=2, pro_a=ops.coef_ptrs(cfd)[0], pro_b=ops.coef_ptrs(cfd)[1], pro_c=ops.coef_ptrs(cfd)[2],
                    pro_cstride=4, in2=x2d, epi_mode=1, out=base.to(dev).to(BF).clone())
    close(o2, F.conv2d(a * x.double() + bb * x2.double() + cc, w.double(), None, padding=1) + base.double())
    u = rb(_rand((N, Cout, H, W), 24) + 0.3)
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1)
    og, tab = ops.conv2d_actbwd(xd, wp, Cout, 3, u.to(dev).to(BF), coef4.to(dev), 0.2, fetch=WG)
    c4 = coef4.double()
    pre = c4[:, 0].view(1, -1, 1, 1) * u.double() + c4[:, 1].view(1, -1, 1, 1)
    refm = F.conv2d(x.double(), w.double(), None, padding=1) * torch.where(pre > 0, 1.0, 0.2)
    safe = (pre.abs() > 1e-4).double()
    close((og.float() * safe.float().to(dev)).to(BF), refm * safe)
    bc = ops.bn_bwd_coefs(tab, 0, coef4.to(dev), N * H * W).cpu().double()
    s1 = refm.sum((0, 2, 3)); s2 = (refm * (u.double() - c4[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
    cnt = N * H * W
    be = -c4[:, 0] * (s2 * c4[:, 3] / cnt) * c4[:, 3]
    ref_bc = torch.stack([c4[:, 0], be, -c4[:, 0] * s1 / cnt - be * c4[:, 2]], 1)
    assert float((bc[:, :3] - ref_bc).abs().max()) < 2e-4 * float(ref_bc.abs().max())


@pytest.mark.parametrize("N,C,H,W", [(16, 16, 256, 256), (16, 64, 320, 320), (16, 128, 160, 160), (16, 256, 40, 40)])
def test_winograd_form_at_full_size_agrees_with_the_direct_form(dev, N, C, H, W):
    """BASELINE configs 2 and 4 at their full sizes (no fp64 reference fits a test's budget there): the two forms compute the same convolution, so they
    must agree to fp32 rounding, element by element (bar: 6e-6 of the output range, 1e-6 rms; measured 1-2e-6 / 1-2e-7), and the form is linear in its input
    (conv(a*x + y) = a*conv(x) + conv(y) to the same bar) - with every prologue / epilogue variant of the loop on the same tensors."""
    from maxstyle_amd import ops
    WG = ops.FETCH_WINOGRAD
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(N, C, H, W, generator=g).to(dev); x2 = torch.randn(N, C, H, W, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev)
    wp = ops.pack_conv_weight(w)
    cf = torch.randn(C, 4, generator=g).to(dev)
    kw = dict(pro_mode=2, pro_a=ops.coef_ptrs(cf)[0], pro_b=ops.coef_ptrs(cf)[1], pro_c=ops.coef_ptrs(cf)[2], pro_cstride=4, in2=x2)

    def agree(a, b):
        rng = float(b.abs().max())
        assert float((a - b).abs().max()) < 6e-6 * rng and float((a - b).pow(2).mean().sqrt()) < 1e-6 * rng, (float((a - b).abs().max()) / rng,)
    d0 = ops.conv2d(x, wp, None, C, 3, 1)
    w0 = ops.conv2d(x, wp, None, C, 3, 1, fetch=WG)
    assert not torch.equal(d0, w0)
    agree(w0, d0)
    agree(ops.conv2d(x, wp, None, C, 3, 1, fetch=WG, **kw), ops.conv2d(x, wp, None, C, 3, 1, **kw))
    u = torch.randn(N, C, H, W, generator=g).to(dev) + 0.3
    c4 = torch.stack([1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g), 0.3 + 0.1 * torch.randn(C, generator=g), 1 + 0.1 * torch.randn(C, generator=g).abs()], 1).to(dev)
    ow, tw = ops.conv2d_actbwd(x, wp, C, 3, u, c4, 0.2, fetch=WG)
    od, td = ops.conv2d_actbwd(x, wp, C, 3, u, c4, 0.2)
    agree(ow, od)
    bw = ops.bn_bwd_coefs(tw, 0, c4, N * H * W); bd = ops.bn_bwd_coefs(td, 0, c4, N * H * W)
    assert float((bw - bd).abs().max()) < 1e-5 * float(bd.abs().max())
    y1 = ops.conv2d(x2, wp, None, C, 3, 1, fetch=WG)
    agree(ops.conv2d(0.5 * x + x2, wp, None, C, 3, 1, fetch=WG), 0.5 * w0 + y1)


def test_inner_loop_with_and_without_the_winograd_form(dev):
    """K = 3 free-running steps at 64x64 on the same networks and style state: the loop that asks for the Winograd form and the one that does not are two
    roundings of the same computation - different bits (the form really runs inside the loop), losses equal to 2e-5, images to 2e-4 of their range."""
    from oracle import maxstyle_oracle as orc
    from maxstyle_amd import engine as E
    from test_engine_gpu import build_engine
    spec = orc.NetSpec(4, 1, 4)
    B, size, layers = 4, 64, [3, 4, 5]
    ew, W, img, lab, styles = build_engine(dev, spec, B, size, layers)
    ed = E.InnerLoopEngine(E.NetSpec(spec.reduce, spec.image_ch, spec.num_classes), B, size, size, dev, lr=0.1)
    ed.winograd = False
    ed.set_nets(ew.nets)
    ed.configure_styles(layers, {i: E.StyleSlot(i, B, spec.channel_num[i]) for i in layers})
    assert ew.winograd and not ed.winograd
    z = ew.encode_fwd(img.to(dev))[0].clone()
    labd = lab.to(dev)
    outs = []
    for eng in (ew, ed):
        for i in layers:
            eng.styles[i].have_std = False
            st = orc.random_style_state(B, spec.channel_num[i], 7 + i)
            eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
        eng.flat_m.zero_(); eng.flat_v.zero_(); eng.flat_g.zero_()
        out = eng.run(z, labd, 3, use_graph=False).float().clone()
        outs.append((out, eng.losses(3).clone()))
        eng.check_errors(sync=True)
    (ow, lw), (od, ld) = outs
    assert not torch.equal(ow, od)
    assert float(((lw - ld) / ld).abs().max()) < 2e-5, (lw, ld)
    assert float((ow - od).abs().max()) < 2e-4 * float(od.abs().max())
