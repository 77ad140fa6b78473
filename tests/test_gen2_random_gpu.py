"""Randomised shapes (seeded) for the second-generation kernels of round 4 against the first-generation kernels they replace (same entry points, process-wide switches) and
against fp64 math: 3x3 stride-2 forward (csrc/ms_conv_s2.h), sub-pixel resampling convs (ms_conv_subpix2.h), LDS-tiled GEMM 1x1 (ms_conv_k1g.h), streaming 1x1 (ms_conv_k1s.h).
The hand-picked shapes of test_conv_gpu.py / test_k1s_gpu.py cover the geometries; these cover what nobody thought of (odd batch sizes, channel counts that are not multiples
of 16, one-row images, images narrower than a tile, block lists that end inside a group)."""
import random

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
    return torch.randn(*shape, generator=g) * scale


def _shapes(seed, n, cin_choices, cout_max, hw_mult, hw_max, nmax=6):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        out.append((rng.randint(1, nmax), rng.choice(cin_choices), rng.randint(1, cout_max), hw_mult[0] * rng.randint(1, hw_max // hw_mult[0]), hw_mult[1] * rng.randint(1, hw_max // hw_mult[1])))
    return out


@pytest.mark.parametrize("N,Cin,Cout,H,W", _shapes(11, 14, [16, 20, 32, 48, 64, 100, 128, 256], 150, (2, 8), 96))
def test_random_stride2(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(x.cpu().double(), w.double(), b.double(), stride=2, padding=1)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)

    def run():
        out = torch.full(tuple(ref.shape), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 3, 2, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d(s2)")
        return out
    was = lib.ms_set_option(b"conv.s2g2", 1)
    try:
        new = run()
        lib.ms_set_option(b"conv.s2g2", 0)
        old = run()
    finally:
        lib.ms_set_option(b"conv.s2g2", was)
    assert rel(new, ref) < 3e-6 and rel(old, ref) < 3e-6 and rel(new, old) < 2e-6


@pytest.mark.parametrize("N,Cin,Cout,Hs,Ws", _shapes(12, 12, [4, 8, 12, 16, 24, 40, 64, 128], 80, (1, 4), 48))
@pytest.mark.parametrize("mode", [0, 1])
def test_random_subpixel(dev, N, Cin, Cout, Hs, Ws, mode):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, Hs, Ws), 1).to(dev)
    outs = {}
    if mode == 0:
        w = _rand((Cout, Cin, 3, 3), 2, 0.1)
        ref = F.conv2d(F.interpolate(x.cpu().double(), scale_factor=2, mode="nearest"), w.double(), padding=1)
        wp = ops.pack_conv_weight(w.to(dev))
        sums = torch.empty(int(lib.ms_subpix_pack_floats(Cin, Cout)), device=dev)
        check(lib.ms_subpix_pack(wp.data_ptr(), sums.data_ptr(), Cin, Cout, st), "ms_subpix_pack")
    else:
        w = _rand((Cin, Cout, 3, 3), 2, 0.1)
        ref = F.conv_transpose2d(x.cpu().double(), w.double(), stride=2, padding=1, output_padding=1)
        wp = ops.pack_conv_weight_dgrad(w.to(dev))
        sums = None
    for flags in (1, 2, 4, 0):
        out = torch.full((N, Cout, 2 * Hs, 2 * Ws), float("nan"), device=dev)
        check(lib.ms_conv_subpix2(x.data_ptr(), out.data_ptr(), wp.data_ptr(), 0 if sums is None else sums.data_ptr(), 0, N, Cin, Hs, Ws, Cout, mode, 0, 0, 0, 0, 1.0, 0, flags, st), "ms_conv_subpix2")
        outs[flags] = out
    assert rel(outs[1], ref) < 3e-6
    for flags in (2, 4, 0):
        assert torch.equal(outs[flags], outs[1]), flags


@pytest.mark.parametrize("N,Cin,Cout,Hs,Ws", [(20, 128, 64, 14, 14), (3, 24, 20, 9, 18), (1, 16, 16, 2, 2), (2, 40, 8, 5, 6), (4, 64, 32, 7, 10), (2, 8, 33, 3, 22)])
def test_subpixel_upsampling_conv_on_even_widths(dev, N, Cin, Cout, Hs, Ws):
    """Mode 0 of the sub-pixel kernel on stored widths that are even but not a multiple of 4 (ms_conv_subpix_eligible == 2: the block geometry with a half block at the end
    of every row; the 14-pixel level of the reference's shipped 224-pixel workload): against fp64 math, against the fused-fetch conv, and its statistics table through
    ms_bn_finalize - bias, channel tails, one-block rows."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ms_conv_subpix_eligible(Hs, Ws) == 2 and lib.ms_conv_subpix_eligible(Hs, Ws + 1) == 0 and lib.ms_conv_subpix_eligible(Hs, Ws + 2) == 1
    x = _rand((N, Cin, Hs, Ws), 1); w = _rand((Cout, Cin, 3, 3), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    xd, bd = x.to(dev), b.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    sums = torch.empty(int(lib.ms_subpix_pack_floats(Cin, Cout)), device=dev)
    check(lib.ms_subpix_pack(wp.data_ptr(), sums.data_ptr(), Cin, Cout, st), "ms_subpix_pack")
    parts = lib.ms_conv_stats_parts(N, 2 * Hs, 2 * Ws)
    outs = []
    for rep in range(2):
        out = torch.full((N, Cout, 2 * Hs, 2 * Ws), float("nan"), device=dev)
        stats = torch.zeros(Cout * parts + 1, 4, device=dev)
        check(lib.ms_conv_subpix2(xd.data_ptr(), out.data_ptr(), wp.data_ptr(), sums.data_ptr(), bd.data_ptr(), N, Cin, Hs, Ws, Cout, 0, stats.data_ptr(), 0, 0, 0, 1.0, 0, 0, st), "ms_conv_subpix2")
        outs.append((out, stats))
    out, stats = outs[0]
    assert torch.equal(out, outs[1][0]) and torch.equal(stats, outs[1][1])
    assert rel(out, ref) < 3e-6
    st2 = torch.zeros_like(stats)
    old = ops.conv2d(xd, wp, bd, Cout, 3, 1, fetch=ops.FETCH_UPS2, stats=st2)
    assert rel(out, old) < 3e-6
    gamma, beta = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    cf1, cf2 = torch.empty(Cout, 4, device=dev), torch.empty(Cout, 4, device=dev)
    check(lib.ms_bn_finalize(stats.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, cf1.data_ptr(), Cout, st), "bn_finalize")
    check(lib.ms_bn_finalize(st2.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, cf2.data_ptr(), Cout, st), "bn_finalize")
    o64 = out.double()
    assert float((cf1[:, 2].double() - o64.mean(dim=(0, 2, 3))).abs().max()) < 2e-6 * float(ref.abs().max())
    assert rel(cf1, cf2) < 3e-5
    # the first generation cannot take these rows: a clear error, not a wrong result
    assert lib.ms_conv_subpix2(xd.data_ptr(), out.data_ptr(), wp.data_ptr(), sums.data_ptr(), bd.data_ptr(), N, Cin, Hs, Ws, Cout, 0, 0, 0, 0, 0, 1.0, 0, 1, st) != 0


@pytest.mark.parametrize("N,Cin,Cout,H,W", _shapes(13, 12, [256, 264, 320, 384, 512], 300, (1, 4), 44, nmax=16))
def test_random_gemm_1x1(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 1, 1), 2, 0.1); b = _rand((Cout,), 3)
    ref = F.conv2d(x.cpu().double(), w.double(), b.double())
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    u = _rand((N, Cout, H, W), 5).to(dev)
    coef = torch.stack([_rand((Cout,), 8).abs() + 0.5, _rand((Cout,), 9), _rand((Cout,), 10) * 0.1, torch.rand(Cout) + 0.5], dim=1).contiguous().to(dev)

    def run():
        a = torch.full((N, Cout, H, W), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, a.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d")
        t = torch.full_like(a, float("nan"))
        check(lib.ms_conv1x1_bnres(x.data_ptr(), t.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), coef.data_ptr(), 0.2, 0, st), "ms_conv1x1_bnres")
        return a, t
    was = lib.ms_set_option(b"conv.k1g", 1)
    try:
        new = run()
        lib.ms_set_option(b"conv.k1g", 0)
        old = run()
    finally:
        lib.ms_set_option(b"conv.k1g", was)
    assert rel(new[0], ref) < 3e-6
    assert torch.equal(new[0], old[0]) and torch.equal(new[1], old[1])


@pytest.mark.parametrize("N,Cin,Cout,H,W", _shapes(14, 10, [64, 128], 200, (2, 2), 130, nmax=16))
def test_random_streaming_1x1(dev, N, Cin, Cout, H, W):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    N = max(N, 8); H = max(H, 64); W = max(W, 64)          # (large enough to be streamed: asserted where it should be)
    x = _rand((N, Cin, H, W), 1).to(dev)
    w = _rand((Cout, Cin, 1, 1), 2, 0.1); b = _rand((Cout,), 3)
    wp = ops.pack_conv_weight(w.to(dev)); bd = b.to(dev)
    u = _rand((N, Cout, H, W), 5).to(dev)
    coef = torch.stack([_rand((Cout,), 8).abs() + 0.5, _rand((Cout,), 9), _rand((Cout,), 10) * 0.1, torch.rand(Cout) + 0.5], dim=1).contiguous().to(dev)

    def run():
        a = torch.full((N, Cout, H, W), float("nan"), device=dev)
        check(lib.ms_conv2d(x.data_ptr(), 0, a.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0, 0, st), "ms_conv2d")
        t = torch.full_like(a, float("nan"))
        check(lib.ms_conv1x1_bnres(x.data_ptr(), t.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, Cin, H, W, Cout, u.data_ptr(), coef.data_ptr(), 0.2, 0, st), "ms_conv1x1_bnres")
        return a, t
    was = lib.ms_set_option(b"conv.k1s", 1)
    try:
        new = run()
        lib.ms_set_option(b"conv.k1s", 0)
        old = run()
    finally:
        lib.ms_set_option(b"conv.k1s", was)
    assert torch.equal(new[0], old[0]) and torch.equal(new[1], old[1])
    ref = F.conv2d(x.cpu().double(), w.double(), b.double())
    assert rel(new[0], ref) < 3e-6
