"""The outer-iteration oracle (oracle/outer_oracle.py) against the golden vectors of the reference's own training-loop body
(tests/golden/make_golden_outer.py; SURVEY 8(f)1,3).  fp64 twin: agreement to round-off; fp32: calibrated (first-step Adam is sign
descent, so single weights whose gradient sign is in the noise move by +-lr either way)."""
import os

import numpy as np
import pytest
import torch

from oracle import maxstyle_oracle as orc
from oracle import outer_oracle as outer


def run_oracle(dtype, B, size, layers, K, n_outer, optimizer="AdamW"):
    spec = orc.NetSpec(4, 1, 4)
    W = orc.procedural_weights(spec, seed=0, dtype=dtype)
    state = outer.new_optimizer_state(W)
    clean, lab = orc.synthetic_batch(B, size, 1, 4, seed=1234)
    clean = clean.to(dtype)
    outs = []
    for it in range(n_outer):
        g = torch.Generator().manual_seed(100 + it)
        noise = (0.05 * torch.randn(clean.shape, generator=g)).to(dtype)
        styles = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i + 10 * it, dtype) for i in layers}
        outs.append(outer.train_iteration(W, state, clean, lab, noise, styles, layers, n_iter=K, lr_inner=0.1, lr_outer=1e-4, optimizer=optimizer))
    return W, outs


def summary(t):
    t = t.detach().double().reshape(-1)
    if t.numel() <= 600:
        return float(t.norm()), float(t.sum()), t.numpy()
    idx = torch.linspace(0, t.numel() - 1, 64).long()
    return float(t.norm()), float(t.sum()), t[idx].numpy()


def golden_vec(d, key):
    return d[key + ".full"] if key + ".full" in d.files else d[key + ".sample"]


def test_outer_oracle_fp64_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "outer_small_f64.npz"))
    W, outs = run_oracle(torch.float64, 4, 64, [3, 4, 5], 2, 2)
    for it, o in enumerate(outs):
        t = f"it{it}."
        got = [o["seg_loss"], o["recon_loss"], o["hard_seg_loss"], o["hard_recon_loss"]]
        assert np.allclose(got, d[t + "losses"], rtol=1e-11, atol=1e-13)
        for k, g in o["grads"].items():
            if outer.is_null_grad_bias(*k.split("/", 1)):
                assert float(g.abs().max()) < 1e-12          # exact value 0; both sides hold round-off
                continue
            n, s, v = summary(g)
            assert abs(n - float(d[t + "grad." + k + ".norm"])) <= 1e-9 * max(n, 1e-30), k
            assert np.allclose(v, golden_vec(d, t + "grad." + k), rtol=1e-7, atol=1e-9 * n), k
    t = "it1."
    for net in outer.NETS:
        for k, v in W[net].items():
            n, s, vec = summary(v)
            assert np.allclose(vec, golden_vec(d, t + f"after.{net}/{k}"), rtol=1e-9, atol=1e-9), (net, k)
    assert int(W["image_encoder"]["general_encoder.inc.1.num_batches_tracked"]) == 2      # only the two clean passes track


def test_outer_oracle_fp32_within_calibrated_noise(golden_dir):
    """fp32 oracle vs the fp64 reference run, beside the fp32 reference's own distance to it."""
    d32, d64 = np.load(os.path.join(golden_dir, "outer_small.npz")), np.load(os.path.join(golden_dir, "outer_small_f64.npz"))
    W, outs = run_oracle(torch.float32, 4, 64, [3, 4, 5], 2, 1)
    o, t = outs[0], "it0."
    got = np.array([o["seg_loss"], o["recon_loss"], o["hard_seg_loss"], o["hard_recon_loss"]])
    assert np.allclose(got, d64[t + "losses"], rtol=2e-4)
    worst = 0.0
    for k, g in o["grads"].items():
        if outer.is_null_grad_bias(*k.split("/", 1)):
            continue
        ref = float(d64[t + "grad." + k + ".norm"])
        noise = abs(float(d32[t + "grad." + k + ".norm"]) - ref) / ref
        err = abs(summary(g)[0] - ref) / ref
        worst = max(worst, err)
        assert err <= max(6 * noise, 2e-2), (k, err, noise)
    # BatchNorm running statistics are a function of the clean forward pass only: tight
    for k in ("general_encoder.inc.1.running_mean", "general_encoder.down2.conv.4.running_var", "code_decoupler.4.running_mean"):
        vec = summary(W["image_encoder"][k])[2]
        assert np.allclose(vec, golden_vec(d64, t + "after.image_encoder/" + k), rtol=1e-4, atol=1e-6), k


def test_outer_oracle_adam_variant(golden_dir):
    d = np.load(os.path.join(golden_dir, "outer_adam.npz"))
    W, outs = run_oracle(torch.float32, 3, 32, [4], 1, 1, optimizer="Adam")
    o = outs[0]
    assert np.allclose([o["seg_loss"], o["recon_loss"], o["hard_seg_loss"], o["hard_recon_loss"]], d["it0.losses"], rtol=2e-4)
    # a weight with a large, sign-stable gradient moves by exactly lr without decay (Adam, step 1)
    k = "segmentation_decoder/final_conv.bias"
    before = orc.procedural_weights(orc.NetSpec(4, 1, 4), 0)["segmentation_decoder"]["final_conv.bias"]
    moved = (W["segmentation_decoder"]["final_conv.bias"] - before).abs()
    assert torch.allclose(moved, torch.full_like(moved, 1e-4), rtol=1e-2)
    assert np.allclose(summary(W["segmentation_decoder"]["final_conv.bias"])[2], golden_vec(d, "it0.after." + k), atol=1e-6)


def test_rescale_intensity_formula():
    x = torch.randn(3, 2, 5, 7) * 4 + 2
    y = outer.rescale_intensity(x)
    assert float(y.reshape(6, -1).min(1).values.abs().max()) == 0.0
    assert torch.allclose(y.reshape(6, -1).max(1).values, torch.ones(6), atol=1e-6)
    const = torch.full((1, 1, 4, 4), 3.0)
    assert float(outer.rescale_intensity(const).abs().max()) == 0.0        # eps keeps a constant plane finite (0/1e-20)
