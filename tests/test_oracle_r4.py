"""Round-4 oracle checks on the CPU: the oracle against the round-4 reference fixtures (tests/golden/make_golden_r4.py - config 4 at its benchmarked size, config 5's
two call shapes with the reference's own random-depth draw), the int8 weight storage of the trained FCN_64, and the bf16 storage emulation hook."""
import os

import numpy as np
import pytest
import torch

from oracle import maxstyle_oracle as orc
from parity_util import rel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NETS = ("image_encoder", "segmentation_decoder", "image_decoder")


def _weights64(dtype):
    z = np.load(os.path.join(GOLDEN, "trained_fcn64_320.npz"))
    W = {n: {} for n in NETS}
    for key in z.files:
        if key.endswith("::scale"):
            continue
        net, k = key.split("/", 1)
        a = z[key]
        if a.dtype == np.int8:
            t = torch.from_numpy(a.astype(np.float32) * z[key + "::scale"][:, None, None, None])      # fp32 product: the value both sides load
        else:
            t = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
        W[net][k] = t.to(dtype) if t.is_floating_point() else t
    return W


def test_trained_fcn64_storage_is_the_network():
    """trained_fcn64_320.npz: conv weights int8 x one fp32 scale per output channel, the rest fp16 (fp32 where fp16 would overflow).  Every tensor of the three
    sub-nets is present with the shapes of the FCN_64 / 3-channel / 2-class network, |q| <= 127, scales positive, nothing non-finite."""
    z = np.load(os.path.join(GOLDEN, "trained_fcn64_320.npz"))
    shapes = orc.param_shapes(orc.NetSpec(1, 3, 2))
    n_int8 = 0
    for net in NETS:
        for k, shp in shapes[net].items():
            a = z[f"{net}/{k}"]
            assert tuple(a.shape) == tuple(shp), (net, k, a.shape, shp)
            if a.dtype == np.int8:
                n_int8 += a.size
                s = z[f"{net}/{k}::scale"]
                assert s.dtype == np.float32 and s.shape == (a.shape[0],) and bool((s > 0).all()) and int(np.abs(a.astype(np.int32)).max()) <= 127
            else:
                assert np.isfinite(a.astype(np.float64)).all()
    assert n_int8 > 20_000_000                     # the 24.5 M-parameter network's convolution weights


def test_oracle_vs_reference_config4_at_size():
    """The fp64 oracle at BASELINE config 4's size (16x3x320x320, trained FCN_64) against the REFERENCE's fp64 run of generate_max_style_image (loop_full_c4.npz):
    the code z_i, the frozen gamma_std / beta_std of every inserted layer and the first loss (one decode + encode + segmentation pass at the injected parameters, no
    update yet) to 1e-9 - the restatement is pinned at the size the GPU test is judged at."""
    g = np.load(os.path.join(GOLDEN, "loop_full_c4.npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    spec = orc.NetSpec(1, 3, 2)
    W = _weights64(torch.float64)
    B, layers = 16, [3, 4, 5]
    img, lab = orc.synthetic_batch(B, 320, 3, 2, seed=1234)
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img.double())
        zf = z_i.reshape(-1)
        idx = torch.linspace(0, zf.numel() - 1, 4096).long()
        assert rel(zf[idx], g["f64.z_i.sample"]) < 1e-9
        styles = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float64) for i in layers}
        recon = orc.apply_max_style(W["image_decoder"], z_i, styles, layers)
        for i in layers:
            assert rel(styles[i].gamma_std, g[f"f64.{i}.gamma_std"]) < 1e-9 and rel(styles[i].beta_std, g[f"f64.{i}.beta_std"]) < 1e-9
        _, z_s = orc.encoder_forward(W["image_encoder"], recon)
        loss = -float(orc.cross_entropy_2d(orc.decoder_forward(W["segmentation_decoder"], z_s, "NN"), lab))
    assert abs(loss - float(g["f64.losses"][0])) < 1e-9 * abs(float(g["f64.losses"][0]))
    # and the fixture's own consistency: the reference's fp32 run is within its stated noise of its fp64 run
    assert float(g["ref_noise.image_max"]) < 5e-3 and float(g["ref_noise.labels_equal"]) > 0.9999
    assert float(g["f64.clean_dice"][0]) > 0.95 and float(g["f64.final_dice"][0]) < 0.7           # a trained network, and a hard example


def test_oracle_vs_reference_config5_acdc_call():
    """The fp32 oracle on config 5's ACDC-shaped call (trained FCN_16 at 16x1x256x256, p = 0.5: the reference's draw under fix_seed applies layers {4, 5} only)
    against the reference's fp32 run (loop_c5_calls.npz): the not-applied layer takes the identity path, the first loss agrees to fp32 forward rounding (3e-5), and
    the bf16-storage oracle's stored distances are reproduced by running the emulation on the first step."""
    g = np.load(os.path.join(GOLDEN, "loop_c5_calls.npz"))
    z = np.load(os.path.join(GOLDEN, "trained_fcn16_256.npz"))
    W = {n: {} for n in NETS}
    for key in z.files:
        net, k = key.split("/", 1)
        a = z[key]
        W[net][k] = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    spec = orc.NetSpec(4, 1, 4)
    B, layers = 16, [3, 4, 5]
    img, lab = orc.synthetic_batch(B, 256, 1, 4, seed=int(g["acdc.seed"]))
    applied = [bool(v) for v in g["acdc.applied"]]
    assert applied == [False, True, True]
    styles = {}
    for i, ap in zip(layers, applied):
        st = orc.random_style_state(B, spec.channel_num[i], 7 + i)
        st.perm = torch.from_numpy(g[f"acdc.{i}.perm"]).clone()
        st.applied = ap
        styles[i] = st
    stb = {i: s.clone() for i, s in styles.items()}                      # (before the fp32 evaluation freezes gamma_std / beta_std in `styles`)
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img)
    _, loss, grads = orc.inner_step_grads(W, z_i, styles, layers, lab)
    assert abs(loss - float(g["acdc.losses"][0])) < 3e-5 * abs(float(g["acdc.losses"][0]))
    assert not any(n.startswith("3.") for n in grads)                     # layer 3 was not applied: no parameters, no gradients
    # the storage emulation: same call, first loss with bf16-rounded stored activations = the fixture's oracle_bf16 first loss
    with orc.stored_as(orc.bf16_store):
        with torch.no_grad():
            zb, _ = orc.encoder_forward(W["image_encoder"], orc.bf16_store(img))
        _, loss_b, _ = orc.inner_step_grads(W, zb, stb, layers, lab)
    assert abs(loss_b - float(g["acdc.oracle_bf16.losses"][0])) < 1e-6 * abs(loss_b)
    assert 1e-5 < abs(loss_b - loss) / abs(loss) < 1e-2                  # bf16 storage moves the first loss by ~5e-4: visible, small


def test_storage_hook_rounds_value_and_gradient_and_restores():
    x = torch.tensor([1.0 + 2.0 ** -9, -3.1415927, 1e-3], requires_grad=True)
    y = orc.bf16_store(x)
    assert torch.equal(y.detach(), x.detach().to(torch.bfloat16).float())
    gin = torch.tensor([1.0 + 2.0 ** -10, 0.3333333, -7.0])
    y.backward(gin)
    assert torch.equal(x.grad, gin.to(torch.bfloat16).float())            # the gradient of a stored tensor is itself stored
    assert orc.STORE is None
    with orc.stored_as(orc.bf16_store):
        assert orc.STORE is orc.bf16_store
        with orc.stored_as(None):
            assert orc.STORE is None
        assert orc.STORE is orc.bf16_store
    assert orc.STORE is None
    # statistics are taken BEFORE the rounding, the rounded copy is what gets normalised (DESIGN.md "bf16 conv stack")
    u = torch.randn(2, 3, 4, 4) * 3 + 1
    w, b = torch.ones(3), torch.zeros(3)
    with orc.stored_as(orc.bf16_store):
        z1 = orc._bn({"n.weight": w, "n.bias": b}, "n", u, "batch")
    m = u.mean(dim=(0, 2, 3), keepdim=True)
    v = ((u - m) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    z2 = (u.to(torch.bfloat16).float() - m) / torch.sqrt(v + orc.BN_EPS)
    assert torch.allclose(z1, z2, atol=1e-6)
