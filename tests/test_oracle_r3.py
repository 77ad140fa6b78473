"""The CPU oracle against the round-3 fixtures produced by the reference (tests/golden/make_golden_r3.py): every non-default argument of
generate_max_style_image (advanced_triplet...py:458-466, maxstyle.py:75-117), the teacher-forced fp64 twins, and the benchmarked size
(16x1x256x256, K=5, trained FCN_16).  CPU only."""
import os

import numpy as np
import pytest
import torch

from parity_util import rel

PN = ("gamma_noise", "beta_noise", "lmda")


def load_weights(golden_dir, name, dtype=torch.float64):
    z = np.load(os.path.join(golden_dir, name))
    W = {"image_encoder": {}, "segmentation_decoder": {}, "image_decoder": {}}
    for key in z.files:
        net, k = key.split("/", 1)
        a = z[key]
        t = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
        W[net][k] = t.to(dtype) if t.is_floating_point() else t
    return W


# what each case of loop_args.npz passes to generate_max_style_image (tests/golden/make_golden_r3.py::ARG_CASES) in the oracle's terms
ARG_SPECS = {
    "nomix": dict(mix_style=False),
    "nonoise": dict(no_noise=True, learn_noise=False),
    "mixfixed": dict(learn_mix=False),
    "noisefixed": dict(learn_noise=False, zero_noise=True),       # noise_learnable=False with no_noise=False: the fixed noise is ZEROS (maxstyle.py:78-80)
    "lw05": dict(loss_weight=0.5),
    "twoterms": dict(loss_weight=0.75, losses_per_step=2),
    "k0": dict(n_iter=0),
    "lr003": dict(lr=0.03, n_iter=2),
    "beta_drawn": dict(learn_noise=False, zero_noise=True, drawn=True),
    "beta_injected": dict(),
}


def arg_case_states(g, case, spec_kw, B, chn, layers, dtype, tag="f64"):
    from oracle import maxstyle_oracle as orc
    styles = {}
    for i in layers:
        st = orc.random_style_state(B, chn[i], 7 + i, dtype)
        st.perm = torch.from_numpy(g[f"{case}.{tag}.{i}.perm"])
        st.applied = bool(g[f"{case}.{tag}.{i}.applied"])
        st.mix_style = spec_kw.get("mix_style", True)
        st.no_noise = spec_kw.get("no_noise", False)
        st.learn_noise = spec_kw.get("learn_noise", True)
        st.learn_mix = spec_kw.get("learn_mix", True)
        if spec_kw.get("zero_noise"):
            st.gamma_noise = torch.zeros_like(st.gamma_noise); st.beta_noise = torch.zeros_like(st.beta_noise)
        if st.mix_style:
            st.lmda = torch.from_numpy(g[f"{case}.initial.{i}.lmda"]).to(dtype)
        else:
            st.lmda = torch.zeros_like(st.lmda)
        styles[i] = st
    return styles


@pytest.mark.parametrize("case", list(ARG_SPECS))
def test_argument_cases_vs_reference_fp64(golden_dir, case):
    from oracle import maxstyle_oracle as orc
    torch.set_num_threads(4)
    g = np.load(os.path.join(golden_dir, "loop_args.npz"))
    kw = ARG_SPECS[case]
    spec = orc.NetSpec(4, 1, 4)
    B, layers = 4, [3, 4, 5]
    W = load_weights(golden_dir, "trained_fcn16.npz")
    img, lab = orc.synthetic_batch(B, 64, 1, 4, seed=777)
    with torch.no_grad():
        z_i = orc.encoder_forward(W["image_encoder"], img.double())[0]
    assert rel(z_i, g["z_i"]) < 1e-5                                           # (the fixture's code is the fp32 run's)
    styles = arg_case_states(g, case, kw, B, spec.channel_num, layers, torch.float64)
    n_iter = kw.get("n_iter", 3)
    tr = orc.InnerLoopTrace()
    out = orc.generate_max_style_image(W, z_i, styles, layers, lab, n_iter=n_iter, lr=kw.get("lr", 0.1), trace=tr, loss_weight=kw.get("loss_weight", 1.0))
    # the reference logs -CE of every 'seg' term (unweighted); the oracle's trace holds the weighted loss
    ref_losses = g[f"{case}.f64.losses"][::kw.get("losses_per_step", 1)]
    got = np.array(tr.losses) / kw.get("loss_weight", 1.0)
    assert len(got) == len(ref_losses)
    if len(got):
        np.testing.assert_allclose(got, ref_losses, rtol=1e-9)
    assert rel(out, g[f"{case}.f64.image"]) < 2e-7                             # (stored as fp32)
    names = [str(n) for n in g[f"{case}.param_names"]]
    learn = orc.style_param_list(styles, layers)[0]
    # the reference's optimiser list holds every nn.Parameter; the ones it never updates (requires_grad False) are not in the oracle's list
    for n in learn:
        assert n in names
    for i in layers:
        for nm in PN:
            ref = g[f"{case}.f64.final.{i}.{nm}"]
            if nm != "lmda" and kw.get("no_noise"):
                continue                                                        # N(0,1) tensors the forward never reads (maxstyle.py:75-77)
            assert rel(getattr(styles[i], nm), ref) < 1e-8 or float(np.abs(ref).max()) == 0.0, (i, nm)
            if f"{i}.{nm}" not in learn and nm == "lmda" and styles[i].mix_style:
                assert np.array_equal(ref.astype(np.float32), g[f"{case}.initial.{i}.lmda"])       # a fixed lmda really stayed fixed in the reference


def test_beta_draws_come_from_the_cpu_generator(golden_dir):
    """always_use_beta=True with noise_learnable=False: randperm, rand(1) and the Beta(0.1, 0.1) sample all come from the CPU generator
    (maxstyle.py:54-61, 105-107), so `fix_seed` pins the whole state - what the GPU product must reproduce (tests/test_round3_gpu.py)."""
    g = np.load(os.path.join(golden_dir, "loop_args.npz"))
    torch.manual_seed(11)
    for i in (3, 4, 5):
        perm = torch.randperm(4)
        while torch.equal(perm, torch.arange(4)):
            perm = torch.randperm(4)
        rand_p = torch.rand(1)
        assert np.array_equal(perm.numpy(), g[f"beta_drawn.f32.{i}.perm"]) and np.array_equal(rand_p.numpy(), g[f"beta_drawn.f32.{i}.rand_p"])
        applied = bool(rand_p < 0.8)
        assert applied == bool(g[f"beta_drawn.f32.{i}.applied"])
        if applied:
            lm = torch.distributions.Beta(0.1, 0.1).sample((4, 1, 1, 1)).float()
            assert np.array_equal(lm.numpy(), g[f"beta_drawn.initial.{i}.lmda"])
    assert [bool(g[f"beta_drawn.f32.{i}.applied"]) for i in (3, 4, 5)].count(True) in (1, 2)       # a strict subset: random depth under fix_seed


@pytest.mark.parametrize("fixture,net,B,layers,K", [("loop_c2small", (4, 1, 4), 4, [3, 4, 5], 5), ("loop_c4small", (1, 3, 2), 4, [3, 4, 5], 3),
                                                    ("loop_all_layers", (4, 1, 4), 3, [0, 1, 2, 3, 4, 5], 2)])
def test_teacher_forced_fp64_twins(golden_dir, fixture, net, B, layers, K):
    """loop_*_tf64.npz: the reference in fp64 for ONE step from the fp32 run's parameters after step s-1 (frozen gamma_std / beta_std of the fp32 run).
    The fp64 oracle from the same point gives the same gradients - which pins both the twins and the oracle's handling of frozen statistics."""
    from oracle import maxstyle_oracle as orc
    torch.set_num_threads(4)
    g = np.load(os.path.join(golden_dir, fixture + ".npz")); t = np.load(os.path.join(golden_dir, fixture + "_tf64.npz"))
    spec = orc.NetSpec(*net)
    W = orc.procedural_weights(spec, 0, torch.float64)
    img, lab = orc.synthetic_batch(B, 64, spec.image_ch, spec.num_classes, seed=1234)
    with torch.no_grad():
        z_i = orc.encoder_forward(W["image_encoder"], img.double())[0]
    for s in range(1, K + 1):
        styles = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float64) for i in layers}
        for i in layers:
            if s > 1:
                for nm in PN:
                    setattr(styles[i], nm, torch.from_numpy(g[f"step{s - 1}.param.{i}.{nm}"]).double())
            styles[i].gamma_std = torch.from_numpy(g[f"{i}.gamma_std"]).double()
            styles[i].beta_std = torch.from_numpy(g[f"{i}.beta_std"]).double()
        _, loss, grads = orc.inner_step_grads(W, z_i, styles, layers, lab)
        assert abs(loss - t["losses"][s - 1]) < 1e-9 * abs(loss)
        for n, gr in grads.items():
            assert rel(gr, t[f"step{s}.grad.{n}"]) < 1e-7, (s, n)


def test_full_size_first_step_vs_reference_fp64(golden_dir):
    """BASELINE config 2 at its real size on the trained networks (loop_full_c2.npz): the fp64 oracle's code, first loss and parameters after
    the first Adam step against the reference's fp64 run (one step: ~20 s of CPU; the K=5 trajectory is the GPU test's business)."""
    from oracle import maxstyle_oracle as orc
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = np.load(os.path.join(golden_dir, "loop_full_c2.npz"))
    spec = orc.NetSpec(4, 1, 4)
    B, layers = 16, [3, 4, 5]
    W = load_weights(golden_dir, "trained_fcn16_256.npz")
    img, lab = orc.synthetic_batch(B, 256, 1, 4, seed=1234)
    with torch.no_grad():
        z_i = orc.encoder_forward(W["image_encoder"], img.double())[0]
    zf = z_i.reshape(-1)
    idx = torch.linspace(0, zf.numel() - 1, 4096).long()
    assert rel(zf[idx], g["f64.z_i.sample"]) < 1e-10
    styles = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float64) for i in layers}
    tr = orc.InnerLoopTrace()
    orc.generate_max_style_image(W, z_i, styles, layers, lab, n_iter=1, lr=0.1, trace=tr)
    assert abs(tr.losses[0] - g["f64.losses"][0]) < 1e-10 * abs(tr.losses[0])
    for i in layers:
        for nm in PN:
            assert rel(getattr(styles[i], nm), g[f"f64.step1.param.{i}.{nm}"]) < 1e-9, (i, nm)
        assert rel(styles[i].gamma_std, g[f"f64.{i}.gamma_std"]) < 1e-10 and rel(styles[i].beta_std, g[f"f64.{i}.beta_std"]) < 1e-10
