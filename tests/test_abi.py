"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h declares
(and nothing in the ctypes table is undeclared), and the host logic of the MaxStyle module mirrors the reference."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "maxstyle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    import maxstyle_amd._lib as L
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (ms_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert L.lib.ms_version() >= 100


def test_every_entry_point_is_tagged_stable_or_internal():
    """the header marks which entry points are the stable operator surface and which are engine-private fusions with preconditions."""
    text = open(os.path.join(ROOT, "include", "maxstyle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decl = re.findall(r"^(MS_STABLE|MS_INTERNAL)?\s*(?:int|size_t|long long|const char\*)\s+(ms_[a-z0-9_]+)\s*\(", text, flags=re.M)
    assert len(decl) == len(_header_symbols())
    untagged = [n for t, n in decl if not t]
    assert not untagged, untagged
    stable = {n for t, n in decl if t == "MS_STABLE"}
    internal = {n for t, n in decl if t == "MS_INTERNAL"}
    assert {"ms_conv2d", "ms_bn_finalize", "ms_style_fwd", "ms_style_bwd", "ms_adam_step", "ms_conv_wgrad", "ms_confusion"} <= stable
    assert {"ms_conv2d_xfin", "ms_conv2d_ride", "ms_step_tail", "ms_head_ce_tail", "ms_conv1x1_bnres_xfin"} <= internal
    assert all(n.replace("_bf16m", "").replace("_bf16", "") in stable for n in stable if n.endswith(("_bf16", "_bf16m")))      # a twin shares its base's tag


def test_missing_extension_fails_loudly(tmp_path, monkeypatch):
    import importlib, maxstyle_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        L._load()


def test_cpu_tensor_is_refused():
    from maxstyle_amd import MaxStyle
    torch.manual_seed(0)
    m = MaxStyle(4, 3, p=1.5, use_gpu=False)
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.randn(4, 3, 8, 8))


def test_maxstyle_host_contract():
    """Constructor/attribute/parameter-order contract of maxstyle.py:14-139 (no GPU needed)."""
    from maxstyle_amd import MaxStyle
    torch.manual_seed(43)
    m = MaxStyle(batch_size=4, num_feature=2, p=1.5, use_gpu=False)
    assert [n for n, _ in m.named_parameters()] == ["gamma_noise", "beta_noise", "lmda"]
    assert m.gamma_noise.shape == (4, 2, 1, 1) and m.lmda.shape == (4, 1, 1, 1)
    assert m.perm.dtype == torch.int64 and not torch.equal(m.perm, torch.arange(4))
    assert m.gamma_std is None and m.beta_std is None and m.data is None
    assert "MaxStyle" in repr(m)
    # identity short-cuts return the very same object (maxstyle.py:146-152)
    x1 = torch.randn(4, 2, 1, 1)
    assert m(x1) is x1
    off = MaxStyle(4, 2, p=-1.0, use_gpu=False)   # rand_p >= p always: "not applied"
    assert list(off.parameters()) == [] and repr(off) == "diffuse style not applied"
    x = torch.randn(4, 2, 3, 3)
    assert off(x) is x
    assert float(off.lmda.abs().sum()) == 0.0
    with pytest.raises(AssertionError, match="turn no_noise=False"):
        MaxStyle(4, 2, p=1.5, no_noise=True, noise_learnable=True, use_gpu=False)
    # reference quirk: a previously applied instance that redraws "not applied" raises TypeError in reset()
    m.p = -1.0
    with pytest.raises(TypeError):
        m.reset()
    # mix_learnable=False keeps lmda a Parameter without grad; mix_style=False makes it a plain zero tensor
    m2 = MaxStyle(4, 2, p=1.5, mix_learnable=False, use_gpu=False)
    assert isinstance(m2.lmda, torch.nn.Parameter) and not m2.lmda.requires_grad
    m3 = MaxStyle(4, 2, p=1.5, mix_style=False, use_gpu=False)
    assert not isinstance(m3.lmda, torch.nn.Parameter) and [n for n, _ in m3.named_parameters()] == ["gamma_noise", "beta_noise"]
    with pytest.raises(ValueError):
        MaxStyle(1, 2, use_gpu=False)     # the reference hangs forever here; we refuse


def test_rng_draw_order_matches_reference_kat(golden_dir):
    """Same torch.manual_seed -> same perm / rand_p / (CPU) noise draws as the reference's __main__ smoke."""
    import numpy as np
    from maxstyle_amd import MaxStyle
    g = np.load(os.path.join(golden_dir, "kat_ramp.npz"))
    torch.manual_seed(43)
    m = MaxStyle(batch_size=4, num_feature=2, p=0.5, use_gpu=False)
    np.testing.assert_array_equal(m.perm.numpy(), g["perm"])
    np.testing.assert_allclose(m.rand_p.numpy(), g["rand_p"])
    np.testing.assert_allclose(m.gamma_noise.detach().numpy(), g["gamma_noise0"])
    np.testing.assert_allclose(m.beta_noise.detach().numpy(), g["beta_noise0"])
    np.testing.assert_allclose(m.lmda.detach().numpy(), g["lmda0"])


def test_state_dict_layout_matches_reference():
    """Module/parameter names of the drop-in networks == the reference's state_dict keys, so .pth files load."""
    import maxstyle_amd as M
    from oracle import maxstyle_oracle as orc
    for spec in (orc.NetSpec(4, 1, 4), orc.NetSpec(1, 3, 2)):
        shapes = orc.param_shapes(spec)
        net = "FCN_16_standard_no_STN" if spec.reduce == 4 else "FCN_64_standard_no_STN"
        S = M.AdvancedTripletReconSegmentationModel(network_type=net, image_ch=spec.image_ch, num_classes=spec.num_classes, use_gpu=False)
        for name, mod in S.model.items():
            sd = mod.state_dict()
            assert set(sd) == set(shapes[name]), (name, set(sd) ^ set(shapes[name]))
            for k, v in sd.items():
                assert tuple(v.shape) == tuple(shapes[name][k]), (name, k)
    assert sum(p.numel() for p in S.model["image_encoder"].parameters()) > 0
    with pytest.raises(NotImplementedError):
        M.AdvancedTripletReconSegmentationModel(network_type="Unet_16", use_gpu=False)


def test_synthetic_matches_oracle():
    """bench.py / tools generate their inputs with maxstyle_amd.synthetic (the product side never imports oracle/): same values as the generators
    the parity tests use, bit for bit."""
    import torch
    from maxstyle_amd import synthetic as syn
    from oracle import maxstyle_oracle as orc
    for net in ((4, 1, 4), (1, 3, 2)):
        a = syn.procedural_weights(syn.NetSpec(*net), 0); b = orc.procedural_weights(orc.NetSpec(*net), 0)
        assert list(a) == list(b)
        for n in a:
            assert list(a[n]) == list(b[n])
            assert all(torch.equal(a[n][k], b[n][k]) for k in a[n])
        assert syn.NetSpec(*net).channel_num == orc.NetSpec(*net).channel_num
    x = syn.synthetic_batch(3, 48, 3, 2, 77); y = orc.synthetic_batch(3, 48, 3, 2, 77)
    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
    s = syn.random_style_state(5, 8, 9); t = orc.random_style_state(5, 8, 9)
    assert all(torch.equal(getattr(s, k), getattr(t, k)) for k in ("perm", "lmda", "gamma_noise", "beta_noise"))


def test_product_never_imports_oracle():
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "maxstyle_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
