"""The encoder's first conv (1 -> 16 channels on the image, encoder_decoder.py:441-445) with the nine taps as the K dimension of the matrix instruction
(csrc/ms_conv_small.hip, conv3x3_k9_kernel; library option "conv.k9"): against fp64 math, against ms_conv2d on the same layer (the SAME BITS in the output), and its statistics table through
ms_bn_finalize - at the benchmarked and the shipped image sizes, ragged heights, one-tile rows, partial strips."""
import pytest
import torch
import torch.nn.functional as F

from parity_util import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


SHAPES = [(16, 256, 256), (20, 192, 192), (20, 224, 224), (2, 5, 48), (1, 1, 16), (3, 7, 80), (16, 320, 320)]


@pytest.mark.parametrize("N,H,W", SHAPES)
def test_taps_as_k_form(dev, N, H, W):
    form = 1
    from maxstyle_amd import ops
    import maxstyle_amd._lib as L
    Cout = 16
    g = torch.Generator().manual_seed(9)
    x = torch.rand(N, 1, H, W, generator=g); w = torch.randn(Cout, 1, 3, 3, generator=g) * 0.3; b = torch.randn(Cout, generator=g) * 0.1
    xd = x.to(dev); wp = ops.pack_conv_weight(w).to(dev); bd = b.to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    parts = L.lib.ms_conv_stats_parts(N, H, W)
    assert L.lib.ms_get_option(b"conv.k9") == 1
    was = L.lib.ms_set_option(b"conv.k9", form)
    try:
        outs = []
        for rep in range(2):
            out = torch.full((N, Cout, H, W), float("nan"), device=dev)
            st = torch.zeros(Cout * parts + 1, 4, device=dev)
            L.check(L.lib.ms_conv3x3_small_cin(xd.data_ptr(), out.data_ptr(), wp.data_ptr(), bd.data_ptr(), N, 1, H, W, Cout, st.data_ptr(), 0), "ms_conv3x3_small_cin")
            outs.append((out, st))
        nb = torch.full((N, Cout, H, W), float("nan"), device=dev)
        L.check(L.lib.ms_conv3x3_small_cin(xd.data_ptr(), nb.data_ptr(), wp.data_ptr(), 0, N, 1, H, W, Cout, 0, 0), "ms_conv3x3_small_cin (no bias, no statistics)")
    finally:
        L.lib.ms_set_option(b"conv.k9", was)
    out, st = outs[0]
    assert torch.equal(out, outs[1][0]) and torch.equal(st, outs[1][1])              # deterministic, fixed-order reductions
    assert rel(out, ref) < 1e-6
    assert rel(nb, F.conv2d(x.double(), w.double(), None, padding=1)) < 1e-6
    st2 = torch.zeros_like(st)
    o2 = ops.conv2d(xd, wp, bd, Cout, 3, 1, stats=st2)
    same = bool(torch.equal(o2, out))
    assert same, float((o2 - out).abs().max())        # the MFMA adds its four K products in K order: the general kernels' tap-by-tap accumulation, bit for bit
    gamma, beta = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    cf1, cf2 = torch.empty(Cout, 4, device=dev), torch.empty(Cout, 4, device=dev)
    L.check(L.lib.ms_bn_finalize(st.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, cf1.data_ptr(), Cout, 0), "bn_finalize")
    L.check(L.lib.ms_bn_finalize(st2.data_ptr(), parts, gamma.data_ptr(), beta.data_ptr(), 1e-5, cf2.data_ptr(), Cout, 0), "bn_finalize")
    o64 = out.double()
    mean64 = o64.mean(dim=(0, 2, 3)); var64 = o64.var(dim=(0, 2, 3), unbiased=False)
    scale = float(ref.abs().max())
    assert float((cf1[:, 2].double() - mean64).abs().max()) < 2e-6 * scale
    assert float((cf1[:, 3].double() - (var64 + 1e-5).rsqrt()).abs().max() / cf1[:, 3].abs().max()) < 2e-5
    assert rel(cf1, cf2) < 2e-5
