"""CPU oracle for the MaxStyle inner adversarial style-optimisation path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is imported by the product package
(`maxstyle_amd/`); only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may use it, and only as the checker / the timed CPU baseline.

This is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the reference algorithm, written from
the specification in SURVEY.md Appendix A and pinned against the reference itself:
  * the reference's only known-answer material (`src/advanced/maxstyle.py:193-241`, values in
    SURVEY.md 8(c)) -> tests/test_oracle.py::test_known_answer_ramp
  * golden vectors produced by importing the reference in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz) -> tests/test_oracle.py

Reference functions restated (all paths relative to /root/reference):
  maxstyle_forward / maxstyle_backward   src/advanced/maxstyle.py:140-189  (+ autograd of it)
  batchnorm_batchstat                    src/models/model_util.py:468-510  (BN in train mode with
                                         track_running_stats=False: batch statistics, frozen affine)
  res_up_block                           src/models/ebm/encoder_decoder.py:289-357
  res_down_block                         src/models/ebm/encoder_decoder.py:22-74
  encoder_forward                        src/models/ebm/encoder_decoder.py:423-482, 634-680
  decoder_forward / apply_max_style      src/models/ebm/encoder_decoder.py:561-631
  cross_entropy_2d                       src/models/custom_loss.py:1043-1078
  adam_step                              torch.optim.Adam defaults
  generate_max_style_image               src/models/advanced_triplet_recon_segmentation_model.py:458-571
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LEAKY = 0.2


# --------------------------------------------------------------------------------------------
# network description + procedural weights (closed-form function of tensor name: both the
# fixture generator and the tests can rebuild them without shipping megabytes of weights)
# --------------------------------------------------------------------------------------------
@dataclass
class NetSpec:
    """FCN_16 (reduce=4) / FCN_64 (reduce=1) of advanced_triplet...py:152-203."""
    reduce: int = 4
    image_ch: int = 1
    num_classes: int = 4

    @property
    def widths(self):  # 64,128,256,512 // reduce
        r = self.reduce
        return [64 // r, 128 // r, 256 // r, 512 // r]

    @property
    def code_ch(self):
        return 512 // self.reduce

    @property
    def channel_num(self):
        # train_adv...py:255-258: channels at layer indexes 0..5 of the image decoder
        r = self.reduce
        return [512 // r, 256 // r, 128 // r, 64 // r, 64 // r, self.image_ch]


def param_shapes(spec: NetSpec) -> Dict[str, Dict[str, tuple]]:
    """state_dict-compatible tensor names/shapes of the three sub-nets."""
    w = spec.widths
    enc: Dict[str, tuple] = {}

    def conv(d, name, cout, cin, k, bias=True):
        d[name + ".weight"] = (cout, cin, k, k)
        if bias:
            d[name + ".bias"] = (cout,)

    def bn(d, name, c):
        d[name + ".weight"] = (c,)
        d[name + ".bias"] = (c,)
        d[name + ".running_mean"] = (c,)
        d[name + ".running_var"] = (c,)
        d[name + ".num_batches_tracked"] = ()

    g = "general_encoder."
    conv(enc, g + "inc.0", w[0], spec.image_ch, 3); bn(enc, g + "inc.1", w[0])
    conv(enc, g + "inc.3", w[0], w[0], 3); bn(enc, g + "inc.4", w[0])
    chans = [(w[0], w[1]), (w[1], w[2]), (w[2], w[3]), (w[3], w[3])]
    for i, (ci, co) in enumerate(chans, start=1):
        p = g + f"down{i}."
        conv(enc, p + "down", ci, ci, 3)
        conv(enc, p + "conv.0", co, ci, 3); bn(enc, p + "conv.1", co)
        conv(enc, p + "conv.3", co, co, 3); bn(enc, p + "conv.4", co)
        conv(enc, p + "conv_input", co, ci, 1)
    conv(enc, g + "final_conv.0", spec.code_ch, w[3], 1); bn(enc, g + "final_conv.1", spec.code_ch)
    conv(enc, "code_decoupler.0", spec.code_ch, spec.code_ch, 3, bias=False); bn(enc, "code_decoupler.1", spec.code_ch)
    conv(enc, "code_decoupler.3", spec.code_ch, spec.code_ch, 3, bias=False); bn(enc, "code_decoupler.4", spec.code_ch)

    def decoder(out_ch, conv_t):
        d: Dict[str, tuple] = {}
        r = spec.reduce
        chans = [(spec.code_ch, 256 // r), (256 // r, 128 // r), (128 // r, 64 // r), (64 // r, 64 // r)]
        for i, (ci, co) in enumerate(chans, start=1):
            p = f"up{i}."
            if conv_t:
                d[p + "up.weight"] = (ci, ci, 2, 2)
                d[p + "up.bias"] = (ci,)
            conv(d, p + "conv.0", co, ci, 3); bn(d, p + "conv.1", co)
            conv(d, p + "conv.3", co, co, 3); bn(d, p + "conv.4", co)
            conv(d, p + "conv_input", co, ci, 1)
        conv(d, "final_conv", out_ch, 64 // r, 1)
        return d

    return {
        "image_encoder": enc,
        "segmentation_decoder": decoder(spec.num_classes, conv_t=False),
        "image_decoder": decoder(spec.image_ch, conv_t=True),
    }


def procedural_weights(spec: NetSpec, seed: int = 0, dtype=torch.float32) -> Dict[str, Dict[str, torch.Tensor]]:
    """Deterministic weights: numpy PCG64 stream seeded by crc32(net/name)+seed.

    Scales follow what is in effect in the reference after get_network: conv weights
    ~ N(0, 2/fan_in) (kaiming), BN gamma ~ N(1, 0.02); unlike the reference init BN beta and conv
    biases are small non-zero values so that every bias path is exercised by the parity tests."""
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for net, shapes in param_shapes(spec).items():
        sd = {}
        for name, shp in shapes.items():
            rng = np.random.Generator(np.random.PCG64(zlib.crc32(f"{net}/{name}".encode()) + seed))
            if name.endswith("num_batches_tracked"):
                t = torch.zeros((), dtype=torch.int64)
            elif name.endswith("running_mean"):
                t = torch.zeros(shp, dtype=dtype)
            elif name.endswith("running_var"):
                t = torch.ones(shp, dtype=dtype)
            elif len(shp) == 4:
                fan_in = shp[1] * shp[2] * shp[3]
                if name.endswith("up.weight"):  # ConvTranspose2d [cin, cout, 2, 2]: fan_in counted as torch does
                    fan_in = shp[1] * shp[2] * shp[3]
                    t = torch.from_numpy(rng.uniform(-1, 1, shp) / np.sqrt(fan_in))
                else:
                    t = torch.from_numpy(rng.standard_normal(shp) * np.sqrt(2.0 / fan_in))
            elif ".conv.1." in name or ".conv.4." in name or "inc.1." in name or "inc.4." in name \
                    or "final_conv.1." in name or "code_decoupler.1." in name or "code_decoupler.4." in name:
                if name.endswith("weight"):
                    t = torch.from_numpy(1.0 + 0.02 * rng.standard_normal(shp))
                else:
                    t = torch.from_numpy(0.02 * rng.standard_normal(shp))
            else:  # conv / convT bias
                t = torch.from_numpy(rng.uniform(-0.05, 0.05, shp))
            # values are DEFINED in fp32 (what a checkpoint would hold); wider dtypes are exact casts of them
            if t.is_floating_point():
                t = t.to(torch.float32).to(dtype)
            sd[name] = t.contiguous()
        out[net] = sd
    return out


# --------------------------------------------------------------------------------------------
# synthetic ACDC-shaped data
# --------------------------------------------------------------------------------------------
def synthetic_batch(batch: int, size: int, image_ch: int = 1, num_classes: int = 4, seed: int = 1234):
    """Images in [0,1] (blobs + low-pass noise, per-slice min-max) and concentric-ellipse labels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.meshgrid(np.linspace(-1, 1, size), np.linspace(-1, 1, size), indexing="ij")
    imgs = np.zeros((batch, image_ch, size, size), np.float32)
    labs = np.zeros((batch, size, size), np.int64)
    for b in range(batch):
        cx, cy = rng.uniform(-0.2, 0.2, 2)
        ax, ay = rng.uniform(0.25, 0.45, 2)
        r = np.sqrt(((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2)
        lab = np.zeros((size, size), np.int64)
        if num_classes >= 4:
            lab[r < 1.0] = 3
            lab[r < 0.7] = 2
            lab[r < 0.4] = 1
        else:
            lab[r < 0.6] = 1
        labs[b] = lab
        for c in range(image_ch):
            img = np.zeros((size, size), np.float64)
            for _ in range(int(rng.integers(3, 7))):
                bx, by = rng.uniform(-0.7, 0.7, 2)
                sx, sy = rng.uniform(0.1, 0.5, 2)
                img += rng.uniform(0.3, 1.0) * np.exp(-(((xx - bx) / sx) ** 2 + ((yy - by) / sy) ** 2))
            img += 0.3 * (lab > 0) + 0.2 * (lab == 2)
            noise = rng.standard_normal((size // 8 + 1, size // 8 + 1))
            noise = np.kron(noise, np.ones((8, 8)))[:size, :size]
            img += 0.1 * noise
            img = (img - img.min()) / (img.max() - img.min() + 1e-20)
            imgs[b, c] = img.astype(np.float32)
    return torch.from_numpy(imgs), torch.from_numpy(labs)


# --------------------------------------------------------------------------------------------
# MaxStyle layer (maxstyle.py:140-189)
# --------------------------------------------------------------------------------------------
@dataclass
class StyleState:
    """Plain-tensor state of one MaxStyle layer (what parity tests *inject*; SURVEY 7 'RNG')."""
    perm: torch.Tensor                     # int64 [B]
    lmda: torch.Tensor                     # [B,1,1,1]
    gamma_noise: torch.Tensor              # [B,C,1,1]
    beta_noise: torch.Tensor               # [B,C,1,1]
    applied: bool = True                   # rand_p < p
    mix_style: bool = True
    no_noise: bool = False
    eps: float = 1e-6
    gamma_std: Optional[torch.Tensor] = None   # [1,C,1,1], frozen after the first forward
    beta_std: Optional[torch.Tensor] = None
    learn_noise: bool = True               # noise_learnable (maxstyle.py:83-98): False -> gamma_noise / beta_noise are plain tensors, never updated
    learn_mix: bool = True                 # mix_learnable (maxstyle.py:113-116): False -> lmda is a Parameter without gradient, skipped by Adam

    def clone(self, dtype=None):
        c = lambda t: None if t is None else t.detach().clone().to(dtype if (dtype and t.is_floating_point()) else t.dtype)
        return StyleState(c(self.perm), c(self.lmda), c(self.gamma_noise), c(self.beta_noise), self.applied,
                          self.mix_style, self.no_noise, self.eps, c(self.gamma_std), c(self.beta_std), self.learn_noise, self.learn_mix)


def random_style_state(batch: int, channels: int, seed: int, dtype=torch.float32, applied=True) -> StyleState:
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(batch, generator=g)
    while batch > 1 and torch.equal(perm, torch.arange(batch)):
        perm = torch.randperm(batch, generator=g)
    return StyleState(
        perm=perm,
        lmda=torch.rand(batch, 1, 1, 1, generator=g).to(dtype),
        gamma_noise=torch.randn(batch, channels, 1, 1, generator=g).to(dtype),
        beta_noise=torch.randn(batch, channels, 1, 1, generator=g).to(dtype),
        applied=applied,
    )


def plane_moments(x: torch.Tensor, eps: float):
    """mu = mean_HW, sig = sqrt(unbiased var_HW + eps)   (maxstyle.py:157-159)."""
    B, C = x.shape[:2]
    xf = x.reshape(B, C, -1)
    n = xf.shape[2]
    mu = xf.mean(dim=2)
    var = ((xf - mu[:, :, None]) ** 2).sum(dim=2) / (n - 1)
    return mu.view(B, C, 1, 1), (var + eps).sqrt().view(B, C, 1, 1)


def batch_std(t: torch.Tensor):
    """unbiased std over the batch dim, keepdim (maxstyle.py:165-168)."""
    B = t.shape[0]
    m = t.mean(dim=0, keepdim=True)
    return (((t - m) ** 2).sum(dim=0, keepdim=True) / (B - 1)).sqrt()


def maxstyle_is_identity(x: torch.Tensor, st: StyleState) -> bool:
    B = x.shape[0]
    hw = x[0, 0].numel()
    return (not st.applied) or (not st.mix_style and st.no_noise) or B <= 1 or hw == 1


def maxstyle_coeffs(mu, sig, st: StyleState):
    """A, S of SURVEY A.1 (per (b,c) affine of x_hat)."""
    if st.mix_style:
        lam = torch.clamp(st.lmda, 0, 1)
        sig_mix = sig * (1 - lam) + sig[st.perm] * lam
        mu_mix = mu * (1 - lam) + mu[st.perm] * lam
    else:
        sig_mix, mu_mix = sig, mu
    if st.no_noise:
        return sig_mix, mu_mix
    return sig_mix + st.gamma_noise * st.gamma_std, mu_mix + st.beta_noise * st.beta_std


def maxstyle_forward(x: torch.Tensor, st: StyleState, return_stats=False):
    """Differentiable w.r.t. x, st.lmda, st.gamma_noise, st.beta_noise (mu/sig detached, as in the reference)."""
    if maxstyle_is_identity(x, st):
        return (x, None, None) if return_stats else x
    assert x.shape[0] == st.perm.shape[0] and x.shape[1] == st.gamma_noise.shape[1], "check input dim"
    with torch.no_grad():
        mu, sig = plane_moments(x, st.eps)
        if st.gamma_std is None:
            st.gamma_std = batch_std(sig)
        if st.beta_std is None:
            st.beta_std = batch_std(mu)
    x_hat = (x - mu) / sig
    A, S = maxstyle_coeffs(mu, sig, st)
    y = A * x_hat + S
    return (y, mu, sig) if return_stats else y


def maxstyle_backward(dy, x, mu, sig, st: StyleState):
    """Closed-form gradients of SURVEY A.2: returns dx, d_gamma_noise, d_beta_noise, d_lmda."""
    x_hat = (x - mu) / sig
    S1 = dy.sum(dim=(2, 3), keepdim=True)
    S2 = (dy * x_hat).sum(dim=(2, 3), keepdim=True)
    A, _ = maxstyle_coeffs(mu, sig, st)
    dx = dy * A / sig
    dgam = st.gamma_std * S2 if not st.no_noise else torch.zeros_like(S2)
    dbet = st.beta_std * S1 if not st.no_noise else torch.zeros_like(S1)
    if st.mix_style:
        inside = ((st.lmda >= 0) & (st.lmda <= 1)).to(dy.dtype)
        dl = inside * ((sig[st.perm] - sig) * S2 + (mu[st.perm] - mu) * S1).sum(dim=1, keepdim=True)
    else:
        dl = torch.zeros_like(st.lmda)
    return dx, dgam, dbet, dl


# --------------------------------------------------------------------------------------------
# conv blocks in "BN batch-stat, frozen affine" mode
# --------------------------------------------------------------------------------------------
def batchnorm_batchstat(u, weight, bias, eps=BN_EPS, u_val=None):
    """u_val: the STORED copy of u that is normalised (statistics always from u itself); None = u."""
    m = u.mean(dim=(0, 2, 3), keepdim=True)
    v = ((u - m) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    return ((u if u_val is None else u_val) - m) / torch.sqrt(v + eps) * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def batchnorm_running(u, weight, bias, rm, rv, eps=BN_EPS):
    """eval-mode BN (running statistics) - used by the Dice/eval row."""
    s = weight / torch.sqrt(rv + eps)
    return u * s.view(1, -1, 1, 1) + (bias - rm * s).view(1, -1, 1, 1)


BN_OBSERVER = None   # optional callable(sd, name, u): oracle/outer_oracle.py uses it for the running-statistics update of a tracking pass

# ---- storage emulation (BASELINE config 5: "bf16 activations") ----------------------------------------------------------------------------------
# STORE: optional callable(tensor) -> tensor applied wherever the HIP engine MATERIALISES an activation (maxstyle_amd/engine.py `a()` tensors): conv outputs
# (after their BatchNorm statistics were taken from the unrounded values - DESIGN.md "bf16 conv stack"), residual-block outputs, up- / down-sampling conv
# outputs, the code, z_i / z_s, MaxStyle outputs and the sigmoid image.  NOT applied to what the engine never writes: BatchNorm-apply + LeakyReLU
# (a conv prologue), the skip-conv sum, the encoder's first block output (`lazy_inc`), the MaxStyle layer in front of the image head (`lazy_style_head`), logits.
# `bf16_store` rounds value AND incoming gradient to bf16 (round-to-nearest-even, fp32 arithmetic in between): the gradient w.r.t. a stored tensor is itself a stored
# tensor in the engine.  The emulation is statistical, not bitwise: it says how large the storage-rounding error of the loop SHOULD be at a given size.
STORE = None


class _Bf16Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def bf16_store(t):
    return _Bf16Store.apply(t)


class stored_as:
    """with stored_as(bf16_store): ... - the oracle's forward / loop with that storage emulation."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global STORE
        self.prev, STORE = STORE, self.fn

    def __exit__(self, *exc):
        global STORE
        STORE = self.prev


def _st(t):
    return t if STORE is None else STORE(t)


def _bn(sd, name, u, bn_mode):
    if BN_OBSERVER is not None:
        BN_OBSERVER(sd, name, u)
    if bn_mode == "batch":
        return batchnorm_batchstat(u, sd[name + ".weight"], sd[name + ".bias"], u_val=(None if STORE is None else STORE(u)))
    return batchnorm_running(_st(u), sd[name + ".weight"], sd[name + ".bias"], sd[name + ".running_mean"], sd[name + ".running_var"])


def _double_conv(sd, p, x, bn_mode, taps=None):
    u1 = F.conv2d(x, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], padding=1)
    a1 = F.leaky_relu(_bn(sd, p + "conv.1", u1, bn_mode), LEAKY)
    u2 = F.conv2d(a1, sd[p + "conv.3.weight"], sd[p + "conv.3.bias"], padding=1)
    z2 = _bn(sd, p + "conv.4", u2, bn_mode)
    if taps is not None:
        taps[p + "u1"] = u1; taps[p + "a1"] = a1; taps[p + "u2"] = u2; taps[p + "z2"] = z2
    return z2


def res_up_block(sd, p, x, up_type, bn_mode="batch", taps=None):
    """encoder_decoder.py:289-357: up -> LeakyReLU(conv1x1(x) + BN(conv3(LReLU(BN(conv3(x))))))."""
    if up_type == "Conv2":
        x = F.conv_transpose2d(x, sd[p + "up.weight"], sd[p + "up.bias"], stride=2)
    elif up_type == "NN":
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    else:
        raise NotImplementedError(up_type)
    if up_type == "Conv2":
        x = _st(x)                             # (nearest up-sampling is fused into the next conv's fetch: nothing is written)
    if taps is not None:
        taps[p + "up"] = x
    s = F.conv2d(x, sd[p + "conv_input.weight"], sd[p + "conv_input.bias"])
    out = _st(F.leaky_relu(s + _double_conv(sd, p, x, bn_mode, taps), LEAKY))
    if taps is not None:
        taps[p + "out"] = out
    return out


def res_down_block(sd, p, x, bn_mode="batch", taps=None):
    """encoder_decoder.py:22-74: conv3 s2 -> LeakyReLU(conv1x1(x) + double_conv(x))."""
    x = _st(F.conv2d(x, sd[p + "down.weight"], sd[p + "down.bias"], stride=2, padding=1))
    if taps is not None:
        taps[p + "down"] = x
    s = F.conv2d(x, sd[p + "conv_input.weight"], sd[p + "conv_input.bias"])
    out = _st(F.leaky_relu(s + _double_conv(sd, p, x, bn_mode, taps), LEAKY))
    if taps is not None:
        taps[p + "out"] = out
    return out


def mixstyle_forward(x, perm=None, lmda=None, gaussian_mu=None, gaussian_std=None, eps=1e-8):
    """MixStyle / DSU with the random draws injected (src/advanced/mixstyle.py:57-108): per-plane mu / sig (unbiased var + eps) are DETACHED;
    'random' / 'crossdomain': statistics interpolated with sample perm[b] by an un-clamped lmda [B,1,1,1]; 'gaussian' (DSU): N(0,1) draws
    [B,C,1,1] scaled by the batch std of mu / sig."""
    mu = x.mean(dim=[2, 3], keepdim=True)
    sig = (x.var(dim=[2, 3], keepdim=True) + eps).sqrt()
    mu, sig = mu.detach(), sig.detach()
    xn = (x - mu) / sig
    if gaussian_mu is not None:
        mu_mix = mu + gaussian_mu * torch.std(mu, dim=0, keepdim=True)
        sig_mix = sig + gaussian_std * torch.std(sig, dim=0, keepdim=True)
    else:
        mu_mix = mu * (1 - lmda) + mu[perm] * lmda
        sig_mix = sig * (1 - lmda) + sig[perm] * lmda
    return xn * sig_mix + mu_mix


def encoder_forward(sd, x, bn_mode="batch", taps=None, mix=None):
    """MyEncoder.forward (+ReLU) then code_decoupler: returns (z_i, z_s). encoder_decoder.py:469-482, 673-680.
    mix: {index 1..6: kwargs of mixstyle_forward} - generate_style_augmented_latent_code (advanced_triplet...py:632-670): MixStyle after inc (1),
    down1..down4 (2..5) and the final activation (6); z_i is then the mixed code."""
    mx = (lambda i, t: mixstyle_forward(t, **mix[i]) if (mix is not None and i in mix) else t)
    g = "general_encoder."
    u = F.conv2d(x, sd[g + "inc.0.weight"], sd[g + "inc.0.bias"], padding=1)
    a = F.leaky_relu(_bn(sd, g + "inc.1", u, bn_mode), LEAKY)
    u = F.conv2d(a, sd[g + "inc.3.weight"], sd[g + "inc.3.bias"], padding=1)
    x1 = F.leaky_relu(_bn(sd, g + "inc.4", u, bn_mode), LEAKY)
    if taps is not None:
        taps[g + "inc.out"] = x1
    h = mx(1, x1)
    for i in range(1, 5):
        h = mx(i + 1, res_down_block(sd, g + f"down{i}.", h, bn_mode, taps))
    u = F.conv2d(h, sd[g + "final_conv.0.weight"], sd[g + "final_conv.0.bias"])
    z_i = mx(6, _st(F.relu(_bn(sd, g + "final_conv.1", u, bn_mode))))
    u = F.conv2d(z_i, sd["code_decoupler.0.weight"], None, padding=1)
    a = F.leaky_relu(_bn(sd, "code_decoupler.1", u, bn_mode), LEAKY)
    u = F.conv2d(a, sd["code_decoupler.3.weight"], None, padding=1)
    z_s = _st(F.relu(_bn(sd, "code_decoupler.4", u, bn_mode)))
    if taps is not None:
        taps["z_i"] = z_i; taps["z_s"] = z_s
    return z_i, z_s


def decoder_forward(sd, code, up_type, last_act=None, bn_mode="batch", taps=None):
    """MyDecoder.forward (encoder_decoder.py:587-596)."""
    h = code
    for i in range(1, 5):
        h = res_up_block(sd, f"up{i}.", h, up_type, bn_mode, taps)
    h = F.conv2d(h, sd["final_conv.weight"], sd["final_conv.bias"])
    if last_act == "sigmoid":
        h = torch.sigmoid(h)
    return h


def apply_max_style(sd, image_code, styles: Dict[int, StyleState], layers: Sequence[int], taps=None, bn_mode="batch"):
    """MyDecoder.apply_max_style for the image decoder ('Conv2' up, Sigmoid) - encoder_decoder.py:598-631.
    bn_mode "running": the sub-networks are in .eval() when the loop is called (nn.BatchNorm2d then normalises with its running statistics
    whatever _disable_tracking_bn_stats toggles: torch/nn/modules/batchnorm.py, `bn_training = self.training or buffers are None`)."""
    x = _st(image_code.detach().clone())
    if 0 in layers:
        x = _st(maxstyle_forward(x, styles[0]))
    for i in range(1, 5):
        x = res_up_block(sd, f"up{i}.", x, "Conv2", bn_mode, taps)
        if i in layers:
            x = maxstyle_forward(x, styles[i])
            if i < 4:
                x = _st(x)                     # (layer 4's output is formed inside the image head: never written)
            if taps is not None:
                taps[f"style{i}"] = x
    x = _st(torch.sigmoid(F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])))
    if 5 in layers:
        x = _st(maxstyle_forward(x, styles[5]))
    return x


def cross_entropy_2d(logits, target):
    """custom_loss.py:1058-1078 with weight=None, mask=None: sum of pixel NLL / (N*H*W)."""
    n, c, h, w = logits.shape
    logp = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    picked = torch.gather(logp, 1, target.view(n, 1, h, w)).sum()
    return -picked / float(n * h * w)


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update (defaults). In-place on p/m/v; t is the 1-based step."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


@dataclass
class InnerLoopTrace:
    losses: List[float] = field(default_factory=list)            # -CE at steps 1..K
    params: List[Dict[str, torch.Tensor]] = field(default_factory=list)   # after each Adam step
    grads: List[Dict[str, torch.Tensor]] = field(default_factory=list)
    images: List[torch.Tensor] = field(default_factory=list)     # recon image after decode i (0..K)


def style_param_list(styles: Dict[int, StyleState], layers: Sequence[int]):
    """Learnable tensors in reference optimiser order: per layer gamma_noise, beta_noise, lmda; layers in list order."""
    names, params = [], []
    for i in layers:
        st = styles[i]
        if not st.applied:
            continue
        for nm in ("gamma_noise", "beta_noise", "lmda"):
            if nm == "lmda" and not (st.mix_style and st.learn_mix):
                continue
            if nm != "lmda" and not st.learn_noise:
                continue
            names.append(f"{i}.{nm}")
            params.append(getattr(st, nm))
    return names, params


def inner_step_grads(weights, image_code, styles, layers, reference_segmentation, bn_mode="batch", loss_weight=1.0):
    """One loss/gradient evaluation of the loop body (advanced_triplet...py:547-561) at the CURRENT style
    parameters: decode -> encode -> segment -> loss = -CE -> d loss / d style params.  Returns
    (recon_image, loss, {name: grad})."""
    enc, seg, dec = weights["image_encoder"], weights["segmentation_decoder"], weights["image_decoder"]
    names, params = style_param_list(styles, layers)
    for k, (n, p) in enumerate(zip(names, params)):
        if not p.requires_grad:
            i, nm = n.split(".")
            p = p.detach().clone().requires_grad_(True)
            setattr(styles[int(i)], nm, p)
            params[k] = p
    recon = apply_max_style(dec, image_code, styles, layers, bn_mode=bn_mode)
    z_i, z_s = encoder_forward(enc, recon, bn_mode)
    logits = decoder_forward(seg, z_s, "NN", None, bn_mode)
    # loss = sum_k w_k * (-CE) over the 'seg' terms (advanced_triplet...py:552-558): loss_weight is their sum
    loss = -cross_entropy_2d(logits, reference_segmentation) * loss_weight if loss_weight != 1.0 else -cross_entropy_2d(logits, reference_segmentation)
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    return recon.detach(), float(loss.detach()), {n: (None if g is None else g.detach()) for n, g in zip(names, grads)}


def generate_max_style_image(weights, image_code, styles: Dict[int, StyleState], layers: Sequence[int],
                             reference_segmentation, n_iter=5, lr=0.1, trace: Optional[InnerLoopTrace] = None,
                             keep_images=False, bn_mode="batch", loss_weight=1.0):
    """The K-step inner loop (advanced_triplet...py:458-571) with *injected* MaxStyle state.

    loss = -CE(seg_decoder(z_s(encoder(recon))), labels); Adam(lr) on the learnable style params in
    reference parameter order.  The image decoded at iteration i is the one the loss of iteration i+1 is
    evaluated on, so `inner_step_grads` (decode+loss+grad at the current params) K times, then one last decode."""
    dec = weights["image_decoder"]
    names, params = style_param_list(styles, layers)
    if n_iter > 0 and len(params) > 0:
        m = {n: torch.zeros_like(p) for n, p in zip(names, params)}
        v = {n: torch.zeros_like(p) for n, p in zip(names, params)}
        steps = {n: 0 for n in names}
        for it in range(n_iter):
            recon, loss, grads = inner_step_grads(weights, image_code, styles, layers, reference_segmentation, bn_mode, loss_weight)
            names, params = style_param_list(styles, layers)
            with torch.no_grad():
                for n, p in zip(names, params):
                    g = grads[n]
                    if g is None:
                        continue
                    steps[n] += 1
                    adam_step(p, g, m[n], v[n], steps[n], lr)
            if trace is not None:
                trace.losses.append(loss)
                trace.grads.append({n: (None if g is None else g.clone()) for n, g in grads.items()})
                trace.params.append({n: p.detach().clone() for n, p in zip(names, params)})
                if keep_images:
                    trace.images.append(recon.clone())
    with torch.no_grad():
        recon = apply_max_style(dec, image_code, styles, layers, bn_mode=bn_mode)
    return recon.detach().clone()


def dice_per_class(pred_labels, target_labels, num_classes):
    """2|A n B| / (|A|+|B|) per foreground class (medpy.metric.binary.dc semantics: 0.0 when both empty)."""
    out = []
    for c in range(1, num_classes):
        a = pred_labels == c
        b = target_labels == c
        denom = int(a.sum()) + int(b.sum())
        out.append(0.0 if denom == 0 else 2.0 * int((a & b).sum()) / denom)
    return out
