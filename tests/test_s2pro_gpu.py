"""3x3 stride-2 forward conv with a BatchNorm-apply + LeakyReLU PROLOGUE on the second-generation kernel (csrc/ms_conv_s2.h, PRO 1; round 5): res_convdown.down of the
first encoder block reads the never-materialised `inc` activation (encoder_decoder.py:40, 423-430).  Against fp64 math and against the first-generation kernel it
replaces (which sums 8-channel chunks tap-major: rounding-level differences, like the prologue-free second generation in test_gen2_random_gpu.py); coefficients from a
table and derived inside the launch (ms_conv2d_xfin)."""
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


CASES = [
    (16, 16, 16, 256, 256),      # config 2's e.d1.xd
    (20, 16, 16, 192, 192),      # shipped ACDC
    (20, 16, 16, 224, 224),      # shipped Prostate (112-pixel outputs: partial tiles)
    (2, 64, 64, 80, 160),        # config 4's widths
    (3, 20, 24, 34, 40),         # ragged: channel tail, partial tiles in both directions
    (1, 16, 40, 2, 8),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", CASES)
def test_stride2_prologue_same_bits_and_fp64(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib
    x = _rand((N, Cin, H, W), 1) * 0.8 + 0.2; w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
    cf = _rand((Cin, 4), 4); cf[:, 1] += 0.4                      # a non-zero shift: lrelu(b) != 0 would leak into the halo if padding came before the prologue
    xa = F.leaky_relu(cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1), 0.2)
    ref = F.conv2d(xa, w.double(), b.double(), stride=2, padding=1)
    wp = ops.pack_conv_weight(w.to(dev)); xd = x.to(dev); cfd = cf.to(dev)
    pa, pb, _ = ops.coef_ptrs(cfd)

    def run():
        return ops.conv2d(xd, wp, b.to(dev), Cout, 3, 2, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
    assert lib.ms_get_option(b"conv.s2g2") == 1
    new = run()
    was = lib.ms_set_option(b"conv.s2g2", 0)
    try:
        old = run()
    finally:
        lib.ms_set_option(b"conv.s2g2", was)
    assert rel(new, ref) < 4e-6 and rel(old, ref) < 4e-6
    print(f"s2 prologue {N}x{Cin}x{H}x{W}->{Cout}: vs fp64 {rel(new, ref):.1e} (first generation {rel(old, ref):.1e}), vs first generation {rel(new, old):.1e}, same bits: {bool(torch.equal(new, old))}")
    assert rel(new, old) < 2e-6
    assert torch.equal(new, run())


def test_stride2_prologue_derived_in_the_launch(dev):
    """ms_conv2d_xfin, stride 2: the launch reduces the producer's statistics table itself - the same output bits and coefficient records as ms_bn_finalize + ms_conv2d,
    over three launch epochs on the same tables."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    N, C, H, W = 16, 16, 128, 128
    x = _rand((N, C, H, W), 1).to(dev); w1 = _rand((C, C, 3, 3), 2, 0.1); w2 = _rand((C, C, 3, 3), 3, 0.1)
    gamma = (1 + 0.1 * _rand((C,), 4)).to(dev); beta = (0.1 * _rand((C,), 5)).to(dev)
    wp1, wp2 = ops.pack_conv_weight(w1.to(dev)), ops.pack_conv_weight(w2.to(dev))
    st = torch.cuda.current_stream().cuda_stream
    stats, parts = ops.conv_stats_buffer(N, C, H, W, dev)
    stats.fill_(0.0)
    gran = torch.zeros(int(lib.ms_xfin_gran_bytes(C)), dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep in range(3):
        u1 = ops.conv2d(x, wp1, None, C, 3, 1, stats=stats)
        coef = ops.bn_finalize(stats, parts, gamma, beta)
        ref = ops.conv2d(u1, wp2, None, C, 3, 2, pro_mode=1, pro_a=ops.coef_ptrs(coef)[0], pro_b=ops.coef_ptrs(coef)[1], pro_cstride=4, slope=0.2)
        out = torch.full_like(ref, float("nan")); coef_x = torch.zeros(C, 4, device=dev)
        check(lib.ms_conv2d_xfin(u1.data_ptr(), 0, out.data_ptr(), wp2.data_ptr(), 0, N, C, H, W, C, 3, 2, 0, 1, 0.2, 0, 0, 0, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 1e-5, 0.0, coef_x.data_ptr(), gran.data_ptr(), err.data_ptr(), st), "ms_conv2d_xfin")
        assert torch.equal(out, ref) and torch.equal(coef_x, coef) and int(err) == 0          # (both through the second-generation kernel: the same bits)
