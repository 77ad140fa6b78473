"""HIP MaxStyle layer (K1/K2/Adam) vs the reference golden vectors and the CPU oracle. Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from parity_util import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _inject(layer, perm, lmda, gn, bn, dev):
    layer.perm = torch.as_tensor(perm).clone()
    layer.rand_p = torch.tensor([0.0])
    with torch.no_grad():
        layer.gamma_noise.data = torch.as_tensor(gn).float().to(dev)
        layer.beta_noise.data = torch.as_tensor(bn).float().to(dev)
        if isinstance(layer.lmda, torch.nn.Parameter):
            layer.lmda.data = torch.as_tensor(lmda).float().to(dev)


@pytest.mark.parametrize("tag", ["a", "b_outside", "c_nomix", "e_big"])
def test_layer_vs_reference_golden(golden_dir, dev, tag):
    from maxstyle_amd import MaxStyle
    g = np.load(os.path.join(golden_dir, "layer_cases.npz"))
    t = lambda k: g[f"{tag}.{k}"]
    x = torch.from_numpy(t("x")).to(dev).requires_grad_(True)
    B, C = x.shape[:2]
    layer = MaxStyle(B, C, p=1.5, mix_style=(tag != "c_nomix"))
    _inject(layer, t("perm"), t("lmda"), t("gamma_noise"), t("beta_noise"), dev)
    y = layer(x)
    y.backward(torch.from_numpy(t("dy")).to(dev))
    assert rel(y, t("y")) < 5e-6
    assert rel(layer.gamma_std, t("gamma_std")) < 5e-6 and rel(layer.beta_std, t("beta_std")) < 5e-6
    assert rel(x.grad, t("dx")) < 2e-5
    assert rel(layer.gamma_noise.grad, t("d_gamma")) < 3e-5
    assert rel(layer.beta_noise.grad, t("d_beta")) < 3e-5
    if tag != "c_nomix":
        assert rel(layer.lmda.grad, t("d_lmda")) < 5e-5
    if tag == "b_outside":
        assert float(layer.lmda.grad[0]) == 0.0 and float(layer.lmda.grad[-1]) == 0.0


def test_known_answer_ramp(golden_dir, dev):
    """The reference's own smoke (maxstyle.py:193-241): 5 Adam steps, losses 4876.38 ... 2869.00."""
    from maxstyle_amd import MaxStyle
    g = np.load(os.path.join(golden_dir, "kat_ramp.npz"))
    feats = torch.from_numpy(g["features"]).to(dev)
    layer = MaxStyle(4, 2, p=1.5)
    _inject(layer, g["perm"], g["lmda0"], g["gamma_noise0"], g["beta_noise0"], dev)
    opt = torch.optim.Adam(list(layer.parameters()), lr=0.1)
    for i in range(5):
        y = layer(feats)
        loss = torch.nn.MSELoss()(y, torch.ones_like(feats))
        opt.zero_grad(); loss.backward(); opt.step()
        assert rel(y, g["outputs"][i]) < 1e-5
        assert abs(loss.item() - g["losses"][i]) < 1e-5 * g["losses"][i]
    assert rel(layer.lmda, g["lmda5"]) < 1e-4 and rel(layer.gamma_noise, g["gamma_noise5"]) < 1e-4
    np.testing.assert_allclose(layer.gamma_std.cpu().numpy(), g["gamma_std"], atol=1e-6)


@pytest.mark.parametrize("shape", [(4, 3, 7, 9), (2, 5, 1, 2), (16, 16, 128, 128), (16, 1, 256, 256), (3, 2, 333, 5), (5, 4, 40, 40)])
def test_layer_vs_oracle_fp64(dev, shape):
    """Seeded inputs, HIP fp32 vs oracle fp64 (ragged / odd / unaligned planes included)."""
    from maxstyle_amd import MaxStyle
    from oracle import maxstyle_oracle as orc
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(shape, generator=gen) * (0.2 + torch.rand(B, C, 1, 1, generator=gen)) + 3.0 * torch.randn(B, C, 1, 1, generator=gen)
    dy = torch.randn(shape, generator=gen)
    st = orc.random_style_state(B, C, 5, torch.float64)
    y64, mu64, sig64 = orc.maxstyle_forward(x.double(), st, return_stats=True)
    dx64, dg64, db64, dl64 = orc.maxstyle_backward(dy.double(), x.double(), mu64, sig64, st)
    layer = MaxStyle(B, C, p=1.5)
    _inject(layer, st.perm, st.lmda.float(), st.gamma_noise.float(), st.beta_noise.float(), dev)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg)
    y.backward(dy.to(dev))
    assert rel(y, y64) < 1e-5
    assert rel(layer._last_stats[0], mu64) < 1e-6 and rel(layer._last_stats[1], sig64) < 1e-5
    assert rel(xg.grad, dx64) < 1e-5
    assert rel(layer.gamma_noise.grad, dg64) < 1e-4
    assert rel(layer.beta_noise.grad, db64) < 1e-4
    assert rel(layer.lmda.grad, dl64) < 1e-4
    # frozen statistics: a second forward on different data reuses gamma_std/beta_std (maxstyle.py:165-168)
    gs = layer.gamma_std.clone()
    layer(xg.detach() * 2 + 1)
    assert torch.equal(gs, layer.gamma_std)


def test_high_mean_low_variance_planes(dev):
    """Post-sigmoid-like planes (mean/sigma ~ 1e3): naive E[x^2]-E[x]^2 would fail here."""
    from maxstyle_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = (100.0 + 0.05 * torch.randn(4, 4, 64, 64, generator=gen))
    mu, sig = ops.style_moments(x.to(dev).contiguous())
    ref = x.double()
    assert rel(mu, ref.mean((2, 3), keepdim=True)) < 1e-6
    assert rel(sig, (ref.var((2, 3), keepdim=True) + 1e-6).sqrt()) < 1e-4


def test_identity_paths_and_errors(dev):
    from maxstyle_amd import MaxStyle
    x = torch.randn(4, 3, 8, 8, device=dev)
    off = MaxStyle(4, 3, p=-1.0)
    assert off(x) is x
    m = MaxStyle(4, 3, p=1.5)
    x1 = torch.randn(4, 3, 1, 1, device=dev)
    assert m(x1) is x1
    with pytest.raises(AssertionError, match="check input dim"):
        m(torch.randn(4, 5, 8, 8, device=dev))
    nn_ = MaxStyle(4, 3, p=1.5, mix_style=False, no_noise=True, noise_learnable=False)
    assert nn_(x) is x


def test_full_size_properties(dev):
    """BASELINE config-2 layer-4 size (16x16x256x256): size-independent invariants of the restyle.
    y has per-plane mean S and per-plane std |A| (x_hat is standardised); permuting the batch with perm=identity-free
    mixing at lmda=0 and zero noise is the identity map."""
    from maxstyle_amd import MaxStyle, ops
    B, C, H, W = 16, 16, 256, 256
    x = torch.randn(B, C, H, W, device=dev) * 0.7 + 0.3
    layer = MaxStyle(B, C, p=1.5)
    y = layer(x)
    mu_y, sig_y = ops.style_moments(y.contiguous())
    mu, sig = layer._last_stats
    lam = layer.lmda.detach().clamp(0, 1)
    A = sig * (1 - lam) + sig[layer.perm.to(dev)] * lam + layer.gamma_noise.detach() * layer.gamma_std
    S = mu * (1 - lam) + mu[layer.perm.to(dev)] * lam + layer.beta_noise.detach() * layer.beta_std
    assert rel(mu_y, S) < 1e-5
    assert rel(sig_y, (A * A + 1e-6).sqrt()) < 1e-4
    with torch.no_grad():
        layer.lmda.zero_(); layer.gamma_noise.zero_(); layer.beta_noise.zero_()
    assert rel(layer(x), x) < 1e-5
    # run-to-run bit equality (deterministic two-stage reductions, no float atomics)
    y1 = layer(x); y2 = layer(x)
    assert torch.equal(y1, y2)


def test_adam_kernel_matches_torch(dev):
    from maxstyle_amd import ops
    gen = torch.Generator().manual_seed(0)
    p0 = torch.randn(1104, generator=gen)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=0.1)
    p = p0.to(dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for t in range(1, 7):
        g = torch.randn(1104, generator=gen) * (10.0 ** (t - 4))
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(dev), m, v, lr=0.1, step=t)
        assert rel(p, p_ref) < 1e-6, t


@pytest.mark.parametrize("shape", [(16, 16, 256, 256), (16, 16, 128, 128), (16, 1, 256, 256), (4, 3, 8, 8), (8, 64, 320, 320), (5, 4, 40, 40)])
def test_fused_single_read_equals_three_kernel(dev, shape):
    """The single-read persistent kernel and the three-launch path agree to rounding; compute_std both ways; repeated launches
    (counter re-initialisation) are bit-identical."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib
    from oracle import maxstyle_oracle as orc
    B, C, H, W = shape
    assert lib.ms_style_fused_ws_bytes(B, C, H * W) > 0
    gen = torch.Generator().manual_seed(1)
    x = (torch.randn(shape, generator=gen) * (0.2 + torch.rand(B, C, 1, 1, generator=gen)) + torch.randn(B, C, 1, 1, generator=gen)).to(dev)
    st = orc.random_style_state(B, C, 9)
    perm = st.perm.to(dev); lm = st.lmda.to(dev).contiguous(); gn = st.gamma_noise.to(dev).contiguous(); bn = st.beta_noise.to(dev).contiguous()
    outs = {}
    for impl in ("3k", "fused"):
        gs = torch.empty(1, C, 1, 1, device=dev); bs = torch.empty(1, C, 1, 1, device=dev)
        y, mu, sig, cA, cS = ops.style_fwd(x, perm, lm, gn, bn, gs, bs, True, impl=impl)
        y2, *_ = ops.style_fwd(x * 1.5 + 0.25, perm, lm, gn, bn, gs, bs, False, impl=impl)      # frozen std on new data
        outs[impl] = (y.clone(), mu.clone(), sig.clone(), cA.clone(), cS.clone(), gs.clone(), bs.clone(), y2.clone())
    for a, b in zip(outs["3k"], outs["fused"]):
        assert rel(b, a) < 2e-6
    y_a = ops.style_fwd(x, perm, lm, gn, bn, outs["fused"][5], outs["fused"][6], False, impl="fused")[0].clone()
    y_b = ops.style_fwd(x, perm, lm, gn, bn, outs["fused"][5], outs["fused"][6], False, impl="fused")[0].clone()
    assert torch.equal(y_a, y_b)
    st = ops.style_ws(B, C, H * W, dev, "fused")            # persistent state of the single-read kernel: [0] launch epoch, [1] error word
    words = st[:8].view(torch.int32)
    assert int(words[1]) == 0, "bounded spin timed out (error word set)"
    assert int(words[0]) == 4, "every launch advances the epoch exactly once"


@pytest.mark.parametrize("tag,mix,lm", [("random", "random", None), ("cross", "crossdomain", None), ("extrap", "random", 1.7), ("dsu", "gaussian", None)])
def test_mixstyle_dsu_vs_reference_golden(golden_dir, dev, tag, mix, lm):
    """MixStyle / DSU baselines (src/advanced/mixstyle.py:44-108) on the K1/K2 kernels; same seed -> same CPU-generator draws as the reference."""
    from maxstyle_amd import MixStyle
    g = np.load(os.path.join(golden_dir, "mixstyle_cases.npz"))
    x = torch.from_numpy(g[f"{tag}.x"]).to(dev).requires_grad_(True)
    layer = MixStyle(p=1.0, alpha=0.1, mix=mix, lmda=lm)
    if mix == "gaussian":
        layer._inject_noise = (torch.from_numpy(g["dsu.gaussian_mu"]), torch.from_numpy(g["dsu.gaussian_std"]))
    torch.manual_seed(5)
    y = layer(x)
    y.backward(torch.from_numpy(g[f"{tag}.dy"]).to(dev))
    if mix != "gaussian":
        np.testing.assert_array_equal(layer.get_perm().numpy(), g[f"{tag}.perm"])
    assert rel(y, g[f"{tag}.y"]) < 1e-5
    assert rel(x.grad, g[f"{tag}.dx"]) < 2e-5
    off = MixStyle(p=-1.0)
    xx = torch.randn(4, 3, 5, 5, device=dev)
    assert off(xx) is xx


@pytest.mark.parametrize("shape", [(16, 16, 256, 256), (16, 16, 128, 128), (16, 1, 256, 256), (4, 8, 16, 24), (6, 5, 8, 8)])
def test_bf16_activation_storage(dev, shape):
    """`*_bf16` entry points (BASELINE config 5): x / y / dy / dx stored as bf16, statistics and arithmetic fp32.
    Tolerance, stated: the statistics equal the fp64 oracle on the bf16-ROUNDED input to fp32 accuracy (1e-5); y and dx are rounded to
    nearest-even bf16 on store, so every element is within 2^-8 relative of the oracle value (2^-9 rounding + fp32 arithmetic), and the parameter
    gradients - accumulated in fp32 from the rounded inputs - within 1e-4."""
    import maxstyle_amd as M
    from oracle import maxstyle_oracle as orc
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(shape, generator=g) * (torch.rand(B, C, 1, 1, generator=g) + 0.2) + torch.randn(B, C, 1, 1, generator=g)).bfloat16()
    dy = torch.randn(shape, generator=g).bfloat16()
    st = orc.random_style_state(B, C, 9, torch.float64)
    y64, mu64, sig64 = orc.maxstyle_forward(x.double(), st, return_stats=True)
    dx64, dg64, db64, dl64 = orc.maxstyle_backward(dy.double(), x.double(), mu64, sig64, st)
    layer = M.MaxStyle(B, C, p=1.5)
    layer.perm = st.perm.clone(); layer.rand_p = torch.tensor([0.0])
    with torch.no_grad():
        layer.gamma_noise.data = st.gamma_noise.float().to(dev); layer.beta_noise.data = st.beta_noise.float().to(dev); layer.lmda.data = st.lmda.float().to(dev)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg)
    assert y.dtype == torch.bfloat16
    y.backward(dy.to(dev))
    mu, sig = layer._last_stats
    assert mu.dtype == torch.float32 and rel(mu.view(B, C), mu64.view(B, C)) < 1e-5 and rel(sig.view(B, C), sig64.view(B, C)) < 1e-5
    tol = 2.0 ** -8
    err_y = ((y.detach().cpu().double() - y64).abs() / (y64.abs() + 1e-3)).max()
    assert float(err_y) < tol, float(err_y)
    assert xg.grad.dtype == torch.bfloat16
    err_dx = ((xg.grad.cpu().double() - dx64).abs() / (dx64.abs() + 1e-3)).max()
    assert float(err_dx) < tol, float(err_dx)
    assert rel(layer.gamma_noise.grad, dg64) < 1e-4 and rel(layer.beta_noise.grad, db64) < 1e-4 and rel(layer.lmda.grad, dl64) < 1e-4
    assert rel(layer.gamma_std, st.gamma_std) < 1e-5 and rel(layer.beta_std, st.beta_std) < 1e-5
