"""Second generation of the 3x3 stride-1 convolution on rows of 12 / 14 / 16 pixels (csrc/ms_conv_k3n.h; library option "conv.k3n"): every prologue / epilogue against
fp64 math on the shapes of the deepest levels of 192 / 224 / 256-pixel inputs (encoder_decoder.py:22-74, 441-445, 650-653), ragged heights, channel tails and partial
M-tiles - and, where the first generation runs 16-channel chunks (rows of 16 and 12 pixels), the SAME BITS in the output tensor as the kernel it replaces."""
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


def _lib():
    from maxstyle_amd._lib import lib
    return lib


CASES = [
    # N, Cin, Cout, H, W
    (16, 128, 128, 16, 16),      # C2's deepest level (two M-tiles per wave)
    (20, 128, 128, 12, 12),      # the reference's shipped ACDC workload (192-pixel crops): 9 M-tiles per image
    (20, 128, 128, 14, 14),      # ... shipped Prostate workload (224): 12.25 M-tiles per image, rows not a multiple of 4 pixels
    (2, 64, 64, 16, 16),         # small grids: one M-tile per wave
    (3, 32, 24, 10, 14),         # ragged height, output-channel tail
    (1, 16, 16, 6, 12),
    (2, 48, 40, 7, 16),          # H*W = 112: the last group of an image is partial
    (5, 16, 33, 2, 14),
]


def _run_all_modes(dev, N, Cin, Cout, H, W):
    """-> dict of outputs of every mode (for the same-bits comparison) after checking each against fp64 math."""
    from maxstyle_amd import ops
    x = _rand((N, Cin, H, W), 11) * 0.7 + 0.3; w = _rand((Cout, Cin, 3, 3), 12, 0.1); b = _rand((Cout,), 13)
    wp = ops.pack_conv_weight(w.to(dev))
    xd = x.to(dev)
    res = {}
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
    stats.fill_(0.0)
    out = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, stats=stats)
    assert rel(out, ref) < 3e-6
    coef = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev))
    assert rel(coef[:, 2], ref.mean((0, 2, 3))) < 1e-5
    assert rel(coef[:, 3], 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-5
    res["plain_stats"] = out; res["plain_stats.coef"] = coef
    res["plain"] = ops.conv2d(xd, wp, None, Cout, 3, 1)
    assert rel(res["plain"], F.conv2d(x.double(), w.double(), None, padding=1)) < 3e-6
    # prologue 1: LeakyReLU(a*x + b) per input channel; zero padding pads the ACTIVATED tensor (a non-zero shift would leak into the halo otherwise)
    cf = _rand((Cin, 4), 14); cf[:, 1] += 0.5
    cfd = cf.to(dev)
    pa, pb, pc = ops.coef_ptrs(cfd)
    xa = F.leaky_relu(cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1), 0.2)
    stats1, _ = ops.conv_stats_buffer(N, Cout, H, W, dev)
    stats1.fill_(0.0)
    o1 = ops.conv2d(xd, wp, b.to(dev), Cout, 3, 1, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2, stats=stats1)
    r1 = F.conv2d(xa, w.double(), b.double(), padding=1)
    assert rel(o1, r1) < 4e-6
    c1 = ops.bn_finalize(stats1, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev))
    assert rel(c1[:, 2], r1.mean((0, 2, 3))) < 1e-5
    res["pro1_stats"] = o1
    # prologue 2: a*x + b*x2 + c (BatchNorm backward of the producer), plain and accumulating epilogues
    x2 = _rand((N, Cin, H, W), 15)
    xb = cf[:, 0].double().view(1, -1, 1, 1) * x.double() + cf[:, 1].double().view(1, -1, 1, 1) * x2.double() + cf[:, 2].double().view(1, -1, 1, 1)
    kw2 = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2.to(dev))
    o2 = ops.conv2d(xd, wp, None, Cout, 3, 1, **kw2)
    r2 = F.conv2d(xb, w.double(), None, padding=1)
    assert rel(o2, r2) < 4e-6
    res["pro2"] = o2
    base = _rand((N, Cout, H, W), 16)
    o3 = ops.conv2d(xd, wp, None, Cout, 3, 1, epi_mode=1, out=base.to(dev).clone(), **kw2)
    assert rel(o3, r2 + base.double()) < 4e-6
    res["pro2_acc"] = o3
    # activation-backward epilogue (ms_conv2d_actbwd) behind both prologues
    u = _rand((N, Cout, H, W), 24) + 0.3
    coef4 = torch.stack([1 + 0.2 * _rand((Cout,), 26), 0.3 * _rand((Cout,), 27), 0.3 + 0.1 * _rand((Cout,), 28), 1 + 0.1 * _rand((Cout,), 29).abs()], 1).to(dev)
    cc = coef4.cpu().double()
    pre = cc[:, 0].view(1, -1, 1, 1) * u.double() + cc[:, 1].view(1, -1, 1, 1)
    safe = pre.abs() > 1e-4
    for tag, kw, rin in (("actbwd", {}, x.double()), ("pro2_actbwd", kw2, xb)):
        gm, tab = ops.conv2d_actbwd(xd, wp, Cout, 3, u.to(dev), coef4, 0.2, **kw)
        rr = F.conv2d(rin, w.double(), None, padding=1) * torch.where(pre > 0, 1.0, 0.2)
        assert rel(gm.cpu().double() * safe, rr * safe) < 4e-6
        bc = ops.bn_bwd_coefs(tab, 0, coef4, N * H * W)
        s1 = rr.sum((0, 2, 3)); s2 = (rr * (u.double() - cc[:, 2].view(1, -1, 1, 1))).sum((0, 2, 3))
        cnt = N * H * W
        be = -cc[:, 0] * (s2 * cc[:, 3] / cnt) * cc[:, 3]
        ref_bc = torch.stack([cc[:, 0], be, -cc[:, 0] * s1 / cnt - be * cc[:, 2]], 1)
        assert rel(bc[:, :3], ref_bc) < 2e-4
        res[tag] = gm; res[tag + ".bc"] = bc
    return res


@pytest.mark.parametrize("N,Cin,Cout,H,W", CASES)
def test_narrow_rows_second_generation_all_modes(dev, N, Cin, Cout, H, W):
    lib = _lib()
    assert lib.ms_get_option(b"conv.k3n") == 1
    new = _run_all_modes(dev, N, Cin, Cout, H, W)
    was = lib.ms_set_option(b"conv.k3n", 0)
    try:
        old = _run_all_modes(dev, N, Cin, Cout, H, W)
    finally:
        lib.ms_set_option(b"conv.k3n", was)
    # where the first generation runs 16-channel chunks (vector staging, one channel block: rows that are a multiple of 4 pixels) the two kernels add the same products
    # in the same order: the same bits in every stored tensor; rows of 14 pixels ran its scalar path with 8-channel chunks: rounding-level differences
    same_bits = (W % 4 == 0)
    for k in ("plain_stats", "plain", "pro1_stats", "pro2", "pro2_acc", "actbwd", "pro2_actbwd"):
        if same_bits:
            assert torch.equal(new[k], old[k]), k
        else:
            assert rel(new[k], old[k]) < 3e-6, k
    for k in ("plain_stats.coef", "actbwd.bc", "pro2_actbwd.bc"):
        assert rel(new[k][:, :3], old[k][:, :3]) < 2e-5, k


def test_second_generation_is_taken_and_deterministic(dev):
    """Two launches give the same bits (fixed-order reductions), and switching the option changes WHICH kernel runs (a different statistics grouping), not what is computed."""
    from maxstyle_amd import ops
    N, C, H, W = 20, 128, 14, 14
    x = _rand((N, C, H, W), 1).to(dev); w = _rand((C, C, 3, 3), 2, 0.05)
    wp = ops.pack_conv_weight(w.to(dev))
    outs = []
    for _ in range(2):
        stats, parts = ops.conv_stats_buffer(N, C, H, W, dev)
        stats.fill_(0.0)
        outs.append((ops.conv2d(x, wp, None, C, 3, 1, stats=stats), stats.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    nslots = int(outs[0][1][0, 0])
    assert torch.equal(outs[0][1][1:], outs[1][1][1:]) and nslots >= 1


def test_cross_workgroup_finalize_prologue_on_narrow_rows(dev):
    """ms_conv2d_xfin (the launch derives the BatchNorm coefficients of its prologue itself) on the second-generation kernel: the same output bits as ms_bn_finalize +
    ms_conv2d, the same coefficient records - kind 0 (BatchNorm apply) and kind 1 (BatchNorm backward from an activation-backward table)."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    N, C, H, W = 16, 64, 16, 16
    x = _rand((N, C, H, W), 1).to(dev); w1 = _rand((C, C, 3, 3), 2, 0.06); w2 = _rand((C, C, 3, 3), 3, 0.06)
    gamma = (1 + 0.1 * _rand((C,), 4)).to(dev); beta = (0.1 * _rand((C,), 5)).to(dev)
    wp1, wp2 = ops.pack_conv_weight(w1.to(dev)), ops.pack_conv_weight(w2.to(dev))
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(3):                                        # launch epochs advance: three rounds on the same tables
        stats, parts = ops.conv_stats_buffer(N, C, H, W, dev) if rep == 0 else (stats, parts)
        if rep == 0:
            stats.fill_(0.0)
            gran = torch.zeros(int(lib.ms_xfin_gran_bytes(C)), dtype=torch.uint8, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
        u1 = ops.conv2d(x, wp1, None, C, 3, 1, stats=stats)
        coef = ops.bn_finalize(stats, parts, gamma, beta)
        ref = ops.conv2d(u1, wp2, None, C, 3, 1, pro_mode=1, pro_a=ops.coef_ptrs(coef)[0], pro_b=ops.coef_ptrs(coef)[1], pro_cstride=4, slope=0.2)
        out = torch.empty_like(ref); coef_x = torch.zeros(C, 4, device=dev)
        check(lib.ms_conv2d_xfin(u1.data_ptr(), 0, out.data_ptr(), wp2.data_ptr(), 0, N, C, H, W, C, 3, 1, 0, 1, 0.2, 0, 0, 0, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 1e-5, 0.0, coef_x.data_ptr(), gran.data_ptr(), err.data_ptr(), st), "ms_conv2d_xfin")
        assert torch.equal(out, ref) and torch.equal(coef_x, coef) and int(err) == 0


@pytest.mark.parametrize("N,Cin,Cout,H", [(20, 128, 128, 14), (3, 64, 40, 10), (1, 80, 16, 2)])
def test_one_by_one_convs_on_14_pixel_rows(dev, N, Cin, Cout, H):
    """The 1x1 convs of the 14-pixel level (final_conv forward with BatchNorm statistics, its data-gradient behind a BatchNorm-backward prologue; encoder_decoder.py:441-445)
    on the narrow-rows kernel (KS = 1 variant: centre tap of the same band) - against fp64 math and against the first-generation kernel they leave (rounding level)."""
    from maxstyle_amd import ops
    W = 14
    lib = _lib()
    x = _rand((N, Cin, H, W), 31) * 0.7 + 0.2; w = _rand((Cout, Cin, 1, 1), 32, 0.1); b = _rand((Cout,), 33)
    x2 = _rand((N, Cin, H, W), 34); cf = _rand((Cin, 4), 35)
    xd, x2d, cfd = x.to(dev), x2.to(dev), cf.to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    pa, pb, pc = ops.coef_ptrs(cfd)
    v = lambda i: cf[:, i].double().view(1, -1, 1, 1)

    def run():
        res = {}
        stats, parts = ops.conv_stats_buffer(N, Cout, H, W, dev)
        stats.fill_(0.0)
        res["stats"] = ops.conv2d(xd, wp, b.to(dev), Cout, 1, 1, stats=stats)
        res["coef"] = ops.bn_finalize(stats, parts, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev))
        res["plain"] = ops.conv2d(xd, wp, None, Cout, 1, 1)
        res["pro1"] = ops.conv2d(xd, wp, None, Cout, 1, 1, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
        res["pro2"] = ops.conv2d(xd, wp, None, Cout, 1, 1, pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2d)
        base = _rand((N, Cout, H, W), 36).to(dev)
        res["pro2_acc"] = ops.conv2d(xd, wp, None, Cout, 1, 1, pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2d, epi_mode=1, out=base.clone())
        res["base"] = base
        return res
    new = run()
    was = lib.ms_set_option(b"conv.k3n", 0)
    try:
        old = run()
    finally:
        lib.ms_set_option(b"conv.k3n", was)
    r0 = F.conv2d(x.double(), w.double(), b.double())
    assert rel(new["stats"], r0) < 3e-6 and rel(new["plain"], F.conv2d(x.double(), w.double())) < 3e-6
    assert rel(new["coef"][:, 2], r0.mean((0, 2, 3))) < 1e-5
    assert rel(new["pro1"], F.conv2d(F.leaky_relu(v(0) * x.double() + v(1), 0.2), w.double())) < 4e-6
    r2 = F.conv2d(v(0) * x.double() + v(1) * x2.double() + v(2), w.double())
    assert rel(new["pro2"], r2) < 4e-6 and rel(new["pro2_acc"], r2 + new["base"].cpu().double()) < 4e-6
    for k in ("stats", "plain", "pro1", "pro2", "pro2_acc"):
        assert rel(new[k], old[k]) < 3e-6, k
    assert rel(new["coef"][:, :3], old["coef"][:, :3]) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,Ho,Wo", [(16, 128, 128, 16, 16), (20, 128, 128, 14, 14), (20, 128, 128, 12, 12), (3, 24, 40, 5, 12), (1, 16, 16, 2, 14), (2, 64, 33, 9, 16)])
def test_stride2_conv_onto_narrow_rows(dev, N, Cin, Cout, Ho, Wo):
    """res_convdown.down of the deepest encoder block (encoder_decoder.py:40: 3x3, stride 2, padding 1) onto rows of 12 / 14 / 16 pixels on the narrow-rows kernel
    (S = 2 variant: the band holds input rows 2 r0 - 1 .. of the flattened output pixels) against fp64 math and the first-generation kernel it leaves."""
    from maxstyle_amd import ops
    lib = _lib()
    x = _rand((N, Cin, 2 * Ho, 2 * Wo), 41); w = _rand((Cout, Cin, 3, 3), 42, 0.08); b = _rand((Cout,), 43)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    xd, wp, bd = x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev)
    run = lambda: ops.conv2d(xd, wp, bd, Cout, 3, 2)
    new = run()
    was = lib.ms_set_option(b"conv.k3n", 0)
    try:
        old = run()
    finally:
        lib.ms_set_option(b"conv.k3n", was)
    assert tuple(new.shape) == (N, Cout, Ho, Wo)
    assert rel(new, ref) < 3e-6 and rel(old, ref) < 3e-6 and rel(new, old) < 3e-6
    assert torch.equal(new, run())
