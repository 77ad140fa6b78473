"""ms_conv2d_actbwd_xfin (round 5): the activation-backward data-gradient conv whose BatchNorm-backward PROLOGUE coefficients are derived inside the launch from the table of
the epilogue that produced its input - against ms_bn_bwd_coefs + ms_conv2d_actbwd: the same bits in the masked gradient and in the coefficient records, the same
BatchNorm-backward coefficients from the new table; on the kernels the encoder's backward takes it on (Winograd wide kernel at the top level, narrow-rows kernel at the
deepest, first generation elsewhere), over three launch epochs on the same tables."""
import pytest
import torch

from parity_util import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("N,C,H,W,wino", [(16, 16, 256, 256, True), (16, 128, 16, 16, False), (20, 128, 14, 14, False), (4, 32, 32, 32, False), (2, 64, 64, 64, True), (3, 24, 10, 20, False)])
def test_actbwd_conv_with_pending_prologue_records(dev, N, C, H, W, wino):
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    fetch = (ops.FETCH_WINOGRAD if wino else 0)
    g0 = _rand((N, C, H, W), 1).to(dev)                      # gradient arriving at layer B's output
    uB = _rand((N, C, H, W), 2).to(dev); uA = _rand((N, C, H, W), 3).to(dev)      # raw BatchNorm inputs of layers B (above) and A (below)
    mk4 = lambda s: torch.stack([1 + 0.2 * _rand((C,), s), 0.3 * _rand((C,), s + 1), 0.1 * _rand((C,), s + 2), 1 + 0.1 * _rand((C,), s + 3).abs()], 1).contiguous().to(dev)
    coefB, coefA = mk4(10), mk4(20)                          # forward records {sc, sh, mean, invstd}
    wB = _rand((C, C, 3, 3), 4, 0.08); wA = _rand((C, C, 3, 3), 5, 0.08)
    dwB, dwA = ops.pack_conv_weight_dgrad(wB.to(dev)), ops.pack_conv_weight_dgrad(wA.to(dev))
    tabB = torch.zeros(lib.ms_conv_actbwd_tab_bytes(C) // 4, device=dev); tabA = torch.zeros_like(tabB); tabA2 = torch.zeros_like(tabB)
    gran = torch.zeros(int(lib.ms_xfin_gran_bytes(C)), dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    cnt = float(N * H * W)
    for rep in range(3):
        # producer: some launch with an activation-backward epilogue fills tabB (here: ms_conv2d_actbwd with a plain prologue-free input)
        gB = torch.empty(N, C, H, W, device=dev)
        check(lib.ms_conv2d_actbwd(g0.data_ptr(), 0, gB.data_ptr(), dwB.data_ptr(), N, C, H, W, C, 3, 1, fetch, 0, 0, 0, 0, 0, 1, 1.0, uB.data_ptr(), coefB.data_ptr(), 0.2,
                                   tabB.data_ptr(), st), "producer")
        # reference: ms_bn_bwd_coefs + ms_conv2d_actbwd
        bc = torch.empty(C, 4, device=dev)
        check(lib.ms_bn_bwd_coefs(tabB.data_ptr(), 0, coefB.data_ptr(), cnt, bc.data_ptr(), C, st), "bn_bwd_coefs")
        pa, pb, pc = ops.coef_ptrs(bc)
        ref = torch.empty(N, C, H, W, device=dev)
        check(lib.ms_conv2d_actbwd(gB.data_ptr(), uB.data_ptr(), ref.data_ptr(), dwA.data_ptr(), N, C, H, W, C, 3, 1, fetch, 2, pa, pb, pc, 0, 4, 1.0, uA.data_ptr(), coefA.data_ptr(),
                                   0.2, tabA.data_ptr(), st), "ms_conv2d_actbwd")
        # one launch
        out = torch.full((N, C, H, W), float("nan"), device=dev); bcx = torch.zeros(C, 4, device=dev)
        check(lib.ms_conv2d_actbwd_xfin(gB.data_ptr(), uB.data_ptr(), out.data_ptr(), dwA.data_ptr(), N, C, H, W, C, 3, 1, fetch, uA.data_ptr(), coefA.data_ptr(), 0.2, tabA2.data_ptr(),
                                        tabB.data_ptr(), coefB.data_ptr(), cnt, bcx.data_ptr(), gran.data_ptr(), err.data_ptr(), st), "ms_conv2d_actbwd_xfin")
        assert int(err) == 0
        assert torch.equal(bcx[:, :3], bc[:, :3])
        assert torch.equal(out, ref)
        b1, b2 = torch.empty(C, 4, device=dev), torch.empty(C, 4, device=dev)
        check(lib.ms_bn_bwd_coefs(tabA.data_ptr(), 0, coefA.data_ptr(), cnt, b1.data_ptr(), C, st), "bn_bwd_coefs")
        check(lib.ms_bn_bwd_coefs(tabA2.data_ptr(), 0, coefA.data_ptr(), cnt, b2.data_ptr(), C, st), "bn_bwd_coefs")
        assert torch.equal(b1, b2)
    # against fp64 math (the last epoch): dx = conv_transpose-free data-gradient of a 3x3 stride-1 conv = conv with flipped weights
    import torch.nn.functional as F
    gin = bc[:, 0].double().cpu().view(1, -1, 1, 1) * gB.double().cpu() + bc[:, 1].double().cpu().view(1, -1, 1, 1) * uB.double().cpu() + bc[:, 2].double().cpu().view(1, -1, 1, 1)
    dx = F.conv_transpose2d(gin, wA.double(), padding=1)
    pre = coefA[:, 0].double().cpu().view(1, -1, 1, 1) * uA.double().cpu() + coefA[:, 1].double().cpu().view(1, -1, 1, 1)
    safe = pre.abs() > 1e-4
    assert rel(out.double().cpu() * safe, dx * torch.where(pre > 0, 1.0, 0.2) * safe) < 5e-6
