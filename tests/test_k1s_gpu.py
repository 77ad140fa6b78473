"""Streaming form of the 1x1 convolutions (csrc/ms_conv_k1s.h) against the tiled first-generation kernel it replaces on large images: same bits for the plain conv,
the residual tail at the same and at twice the resolution (encoder_decoder.py:62-64, 344-346), with a rider job and with the cross-workgroup finalize; and against
fp64 math."""
import pytest
import torch
import torch.nn.functional as F

from parity_util import rel

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# streamed shapes (asserted below): unit counts with a ragged last unit (H W % 64 != 0), several units per wave, 1 - 4 channel blocks, both channel counts; and two
# shapes that stay on the tiled kernel (the switch must be a no-op there)
SHAPES = [(16, 64, 64, 64, 64), (16, 64, 64, 160, 160), (8, 128, 256, 64, 48), (16, 64, 128, 48, 48), (16, 64, 64, 66, 62), (16, 128, 64, 90, 92), (3, 32, 16, 100, 84), (16, 16, 16, 64, 64)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture
def k1s():
    from maxstyle_amd._lib import lib
    was = lib.ms_set_option(b"conv.k1s", 1)
    yield lib
    lib.ms_set_option(b"conv.k1s", was)


@pytest.mark.parametrize("N,Cin,Cout,H,W", SHAPES)
def test_streaming_1x1_same_bits_as_tiled(dev, k1s, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import check
    lib = k1s
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 1, 1), 2, 0.2); b = _rand((Cout,), 3)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    ref = F.conv2d(x.cpu().double(), w.double(), b.double())

    def plain():
        out = torch.full((N, Cout, H, W), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d")
        return out
    u = _rand((N, Cout, H, W), 5).to(dev)
    u2 = _rand((N, Cout, 2 * H, 2 * W), 6).to(dev)
    coef = torch.stack([_rand((Cout,), 8).abs() + 0.5, _rand((Cout,), 9), _rand((Cout,), 10) * 0.1, torch.rand(Cout) + 0.5], dim=1).contiguous().to(dev)

    def tail(up2):
        uu = u2 if up2 else u
        out = torch.full_like(uu, float("nan"))
        check(lib.ms_conv1x1_bnres(x.data_ptr(), out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, uu.data_ptr(), coef.data_ptr(), 0.2, up2, st), "ms_conv1x1_bnres")
        return out
    if Cin >= 64:
        assert lib.ms_conv_k1s_would_run(N, Cin, H, W, Cout, 0) == 1 and lib.ms_conv_k1s_would_run(N, Cin, H, W, Cout, 4) == 1, "this shape is meant to be streamed"
    lib.ms_set_option(b"conv.k1s", 1)
    a0, a1, a2 = plain(), tail(0), tail(1)
    lib.ms_set_option(b"conv.k1s", 0)
    b0, b1, b2 = plain(), tail(0), tail(1)
    lib.ms_set_option(b"conv.k1s", 1)
    assert rel(a0, ref) < 3e-6
    assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(a2, b2)
    # the residual tail against its definition
    sk = ref + 0.0
    t1 = F.leaky_relu(coef[:, 0].cpu().double().view(1, -1, 1, 1) * u.cpu().double() + coef[:, 1].cpu().double().view(1, -1, 1, 1) + sk, 0.2)
    assert rel(a1, t1) < 3e-6
    t2 = F.leaky_relu(coef[:, 0].cpu().double().view(1, -1, 1, 1) * u2.cpu().double() + coef[:, 1].cpu().double().view(1, -1, 1, 1) + F.interpolate(sk, scale_factor=2, mode="nearest"), 0.2)
    assert rel(a2, t2) < 3e-6


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(16, 64, 64, 128, 128), (8, 64, 128, 80, 80), (16, 16, 16, 128, 128)])
def test_streaming_1x1_rider_and_cross_workgroup_finalize(dev, k1s, N, Cin, Cout, H, W):
    """The side jobs travel on the streaming kernel as they do on the tiled one: the rider's records (ms_bn_bwd_coefs / ms_bn_finalize arithmetic) and the residual tail
    whose BatchNorm coefficients are derived inside the launch from the statistics table of the conv in front (`_xfin`): same bits as the separate launches."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import check
    lib = k1s
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 1, 1), 2, 0.2); b = _rand((Cout,), 3)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    # a statistics table: from a 3x3 conv with Cout channels at the tail's resolution
    w3 = _rand((Cout, Cin, 3, 3), 4, 0.1)
    wp3 = ops.pack_conv_weight(w3.to(dev))
    u = torch.empty(N, Cout, H, W, device=dev)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    check(lib.ms_conv2d(x.data_ptr(), 0, u.data_ptr(), wp3.data_ptr(), 0, N, Cin, H, W, Cout, 3, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, stats.data_ptr(), st), "ms_conv2d(3x3)")
    gamma, beta = (_rand((Cout,), 11).abs() + 0.5).to(dev), _rand((Cout,), 12).to(dev)
    coef = torch.empty(Cout, 4, device=dev)
    check(lib.ms_bn_finalize(stats.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, coef.data_ptr(), Cout, st), "ms_bn_finalize")
    want = torch.empty_like(u)
    check(lib.ms_conv1x1_bnres(x.data_ptr(), want.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), coef.data_ptr(), 0.2, 0, st), "ms_conv1x1_bnres")
    got = torch.full_like(u, float("nan"))
    coef2 = torch.zeros(Cout, 4, device=dev)
    gran = torch.zeros(int(lib.ms_xfin_gran_bytes(Cout)), dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.ms_conv1x1_bnres_xfin(x.data_ptr(), got.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                    coef2.data_ptr(), gran.data_ptr(), err.data_ptr(), 0.2, 0, st), "ms_conv1x1_bnres_xfin")
    assert int(err.item()) == 0
    assert torch.equal(coef2, coef) and torch.equal(got, want)
    # rider kind 1 (ms_bn_finalize job) on a plain 1x1 conv
    out4 = torch.zeros(Cout, 4, device=dev)
    o1 = torch.empty(N, Cout, H, W, device=dev)
    check(lib.ms_conv2d_ride(x.data_ptr(), 0, o1.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0,
                             1, stats.data_ptr(), 0, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.0, out4.data_ptr(), Cout, st), "ms_conv2d_ride")
    o0 = torch.empty_like(o1)
    check(lib.ms_conv2d(x.data_ptr(), 0, o0.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d")
    assert torch.equal(o1, o0) and torch.equal(out4, coef)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(16, 64, 64, 80, 80), (4, 128, 32, 96, 100), (16, 64, 16, 64, 64)])
def test_streaming_1x1_conv_transpose_gemm_same_bits(dev, k1s, N, Cin, Cout, H, W):
    """nn.ConvTranspose2d(k=2, s=2) as a GEMM with 4 Cout columns and the pixel-shuffle epilogue (res_up_family 'Conv2' up-sampling, encoder_decoder.py:301-303):
    streaming kernel == tiled kernel bit for bit, == fp64 math."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import check
    lib = k1s
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cin, Cout, 2, 2), 2, 0.2); b = _rand((Cout,), 3)
    assert lib.ms_conv_k1s_would_run(N, Cin, H, W, Cout, 2) == 1
    wp = ops.pack_convT_weight(w.to(dev)); bd = b.to(dev)
    ref = F.conv_transpose2d(x.cpu().double(), w.double(), b.double(), stride=2)

    def run():
        out = torch.full((N, Cout, 2 * H, 2 * W), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 2, 0, st), "ms_conv2d(epi 2)")
        return out
    lib.ms_set_option(b"conv.k1s", 1)
    a = run()
    lib.ms_set_option(b"conv.k1s", 0)
    t = run()
    lib.ms_set_option(b"conv.k1s", 1)
    assert rel(a, ref) < 3e-6
    assert torch.equal(a, t)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(16, 512, 512, 20, 20), (16, 256, 512, 40, 40), (3, 256, 80, 18, 22), (16, 512, 256, 20, 20), (2, 320, 48, 64, 64), (16, 256, 16, 8, 8),
                                               (20, 128, 128, 14, 14), (20, 128, 64, 14, 14), (3, 64, 40, 10, 14)])      # rows of 14 pixels: from 64 channels (round 5)
def test_lds_tiled_1x1_gemm_same_bits_as_tiled(dev, N, Cin, Cout, H, W):
    """LDS-tiled GEMM form of the channel-heavy 1x1 convolutions (csrc/ms_conv_k1g.h) against the tiled kernel: same bits for the plain conv and the residual tail, the
    rider's records and the in-launch BatchNorm finalize of the tail unchanged; against fp64 math.  Shapes: ragged units (H W % 64 != 0), channel counts that are not
    multiples of 16 / 64, 1 / 2 / 4 channel blocks per tile, fewer work items than CUs."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 1, 1), 2, 0.1); b = _rand((Cout,), 3)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    ref = F.conv2d(x.cpu().double(), w.double(), b.double())
    u = _rand((N, Cout, H, W), 5).to(dev)
    coef = torch.stack([_rand((Cout,), 8).abs() + 0.5, _rand((Cout,), 9), _rand((Cout,), 10) * 0.1, torch.rand(Cout) + 0.5], dim=1).contiguous().to(dev)
    # statistics table for the `_xfin` tail: from a 3x3 conv at the same resolution
    w3 = _rand((Cout, 16, 3, 3), 4, 0.1)
    x3 = _rand((N, 16, H, W), 6).to(dev)
    u3 = torch.empty(N, Cout, H, W, device=dev)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    check(lib.ms_conv2d(x3.data_ptr(), 0, u3.data_ptr(), ops.pack_conv_weight(w3.to(dev)).data_ptr(), 0, N, 16, H, W, Cout, 3, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, stats.data_ptr(), st), "conv3x3")
    gamma, beta = (_rand((Cout,), 11).abs() + 0.5).to(dev), _rand((Cout,), 12).to(dev)

    def run_all():
        plain = torch.full((N, Cout, H, W), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, plain.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d")
        tail = torch.full_like(plain, float("nan"))
        check(lib.ms_conv1x1_bnres(x.data_ptr(), tail.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), coef.data_ptr(), 0.2, 0, st), "ms_conv1x1_bnres")
        xt = torch.full_like(plain, float("nan"))
        coef2 = torch.zeros(Cout, 4, device=dev)
        gran = torch.zeros(int(lib.ms_xfin_gran_bytes(Cout)), dtype=torch.uint8, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.ms_conv1x1_bnres_xfin(x.data_ptr(), xt.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u3.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                        coef2.data_ptr(), gran.data_ptr(), err.data_ptr(), 0.2, 0, st), "ms_conv1x1_bnres_xfin")
        assert int(err.item()) == 0
        out4 = torch.zeros(Cout, 4, device=dev)
        rid = torch.full_like(plain, float("nan"))
        check(lib.ms_conv2d_ride(x.data_ptr(), 0, rid.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0,
                                 1, stats.data_ptr(), 0, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.0, out4.data_ptr(), Cout, st), "ms_conv2d_ride")
        u2 = _rand((N, Cout, 2 * H, 2 * W), 13).to(dev)
        up = torch.full_like(u2, float("nan"))
        check(lib.ms_conv1x1_bnres(x.data_ptr(), up.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u2.data_ptr(), coef.data_ptr(), 0.2, 1, st), "ms_conv1x1_bnres(up2)")
        return plain, tail, xt, coef2, rid, out4, up
    was = lib.ms_set_option(b"conv.k1g", 1)
    try:
        new = run_all()
        lib.ms_set_option(b"conv.k1g", 0)
        old = run_all()
    finally:
        lib.ms_set_option(b"conv.k1g", was)
    assert rel(new[0], ref) < 3e-6
    for a_, b_, what in zip(new, old, ("plain", "tail", "xfin tail", "xfin records", "rider conv", "rider records", "half-resolution tail")):
        assert torch.equal(a_, b_), what
