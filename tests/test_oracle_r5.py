"""Round-5 oracle checks on the CPU: the fp64 oracle at the reference's SHIPPED call shapes (config/ACDC/1500_epoch/MICCAI2022_MaxStyle.json: 20x1x192x192, 4 classes;
config/Prostate/MICCAI2022_MaxStyle.json: 20x1x224x224, 2 classes, always_use_beta) against the reference's fp64 runs of exactly those calls
(tests/golden/loop_shipped_acdc.npz / loop_shipped_prostate.npz, make_golden_r5.py), and the internal consistency of the round's fixtures (draws, teacher-forced keys)."""
import os

import numpy as np
import pytest
import torch

from oracle import maxstyle_oracle as orc
from parity_util import rel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NETS = ("image_encoder", "segmentation_decoder", "image_decoder")
CALLS = {"acdc": dict(fixture="loop_shipped_acdc.npz", weights="trained_fcn16_192.npz", size=192, net=(4, 1, 4)),
         "prostate": dict(fixture="loop_shipped_prostate.npz", weights="trained_fcn16_p224.npz", size=224, net=(4, 1, 2))}


def _weights(name, dtype):
    z = np.load(os.path.join(GOLDEN, name))
    W = {n: {} for n in NETS}
    for key in z.files:
        net, k = key.split("/", 1)
        a = z[key]
        t = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
        W[net][k] = t.to(dtype) if t.is_floating_point() else t
    return W


@pytest.mark.parametrize("which", ["acdc", "prostate"])
def test_oracle_vs_reference_shipped_call(which):
    """z_i, the frozen gamma_std / beta_std of every inserted layer and the first loss of the shipped call (one decode + encode + segmentation pass at the injected
    parameters, `initial.{i}.lmda` = the Beta(0.1, 0.1) draws of the Prostate call) to 1e-9 of the reference's fp64 run: the restatement is pinned on the 12- / 14- /
    24- / 28-pixel levels the benchmarked configurations never reach."""
    c = CALLS[which]
    g = np.load(os.path.join(GOLDEN, c["fixture"]))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    spec = orc.NetSpec(*c["net"])
    W = _weights(c["weights"], torch.float64)
    B, layers, K = int(g["B"]), [int(i) for i in g["layers"]], int(g["K"])
    assert (B, layers, K, int(g["size"])) == (20, [3, 4, 5], 5, c["size"])
    img, lab = orc.synthetic_batch(B, c["size"], spec.image_ch, spec.num_classes, seed=1234)
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img.double())
        zf = z_i.reshape(-1)
        idx = torch.linspace(0, zf.numel() - 1, 4096).long()
        assert rel(zf[idx], g["f64.z_i.sample"]) < 1e-9
        styles = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float64) for i in layers}
        for i in layers:
            styles[i].lmda = torch.from_numpy(g[f"initial.{i}.lmda"]).double()
        recon = orc.apply_max_style(W["image_decoder"], z_i, styles, layers)
        for i in layers:
            assert rel(styles[i].gamma_std, g[f"f64.{i}.gamma_std"]) < 1e-9 and rel(styles[i].beta_std, g[f"f64.{i}.beta_std"]) < 1e-9
        _, z_s = orc.encoder_forward(W["image_encoder"], recon)
        loss = -float(orc.cross_entropy_2d(orc.decoder_forward(W["segmentation_decoder"], z_s, "NN"), lab))
    assert abs(loss - float(g["f64.losses"][0])) < 1e-9 * abs(float(g["f64.losses"][0]))
    assert min(float(v) for v in g["f64.clean_dice"]) > 0.9 and max(float(v) for v in g["f64.final_dice"]) < 0.4        # a trained network, and a hard example


@pytest.mark.parametrize("which", ["acdc", "prostate"])
def test_shipped_fixture_is_self_consistent(which):
    """What the GPU bars read from the fixture: three fp32 evaluations of the reference per call (draw 0 = the fixture's own fp32 leg), per-sample errors, and the
    teacher-forced keys - step-k loss equal to the free-running fp64 run's (the injection reproduces the run's state), gradients for every step and tensor."""
    g = np.load(os.path.join(GOLDEN, CALLS[which]["fixture"]))
    K, names = int(g["K"]), [str(n) for n in g["ref_draws.tensor_names"]]
    assert len(names) == 9 and [str(v) for v in g["ref_draws.variants"]] == ["mkldnn_t8", "mkldnn_t2", "native_t8"]
    assert g["ref_draws.image_rms_per_sample"].shape == (3, 20) and g["ref_draws.step1_grad_err"].shape == (3, 9) and g["ref_draws.losses_rel"].shape == (3, K)
    assert abs(float(g["ref_draws.image_max"][0]) - float(g["ref_noise.image_max"])) <= 1e-2 * float(g["ref_noise.image_max"])
    assert float(np.max(g["ref_draws.image_max"])) < 2e-3 and float(np.min(g["ref_draws.labels_equal"])) > 0.9998
    for k in range(2, K + 1):
        assert abs(float(g[f"forced.f64.step{k}.loss"]) - float(g["f64.losses"][k - 1])) <= 1e-9 * abs(float(g["f64.losses"][k - 1]))
        for n in names:
            gr = g[f"forced.f64.step{k}.grad.{n}"]
            assert gr.dtype == np.float64 and gr.shape == g[f"f64.step1.grad.{n}"].shape and np.isfinite(gr).all() and float(np.abs(gr).max()) > 0
    assert g["forced.draws.grad_err"].shape == (2, K - 1, 9) and float(g["forced.draws.grad_err"].max()) < 2e-2 and float(g["forced.draws.loss_rel"].max()) < 3e-6


def test_forced_full_and_draws_fixtures_are_self_consistent():
    f = np.load(os.path.join(GOLDEN, "loop_forced_full.npz"))
    for tag, fixture, K in (("c2", "loop_full_c2.npz", 5), ("c4", "loop_full_c4.npz", 10)):
        g = np.load(os.path.join(GOLDEN, fixture))
        assert int(f[f"{tag}.steps_done"]) == K == int(g["K"])
        for k in range(1, K + 1):
            assert abs(float(f[f"{tag}.step{k}.loss"]) - float(g["f64.losses"][k - 1])) <= 1e-9 * abs(float(g["f64.losses"][k - 1]))
        assert f[f"{tag}.draws.grad_err"].shape == (2, K, 9) and float(f[f"{tag}.draws.grad_err"].max()) < 2e-2 and float(f[f"{tag}.draws.loss_rel"].max()) < 3e-6
    d = np.load(os.path.join(GOLDEN, "loop_ref_draws.npz"))
    assert [str(v) for v in d["variants"]] == ["mkldnn_t8", "mkldnn_t2", "native_t8"]
    for tag, K in (("c4", 10), ("acdc", 5), ("prostate", 10)):
        assert d[f"{tag}.losses_rel"].shape == (3, K) and d[f"{tag}.strided_rms"].shape == (3,) and float(d[f"{tag}.strided_rms"].max()) < 5e-4 and float(d[f"{tag}.labels_equal"].min()) > 0.9999
