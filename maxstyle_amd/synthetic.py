"""Synthetic inputs for benchmarks, smoke runs and profiling tools: procedurally filled weights of the three sub-networks (state_dict names and
shapes of the reference's FCN_16 / FCN_64 solver: advanced_triplet_recon_segmentation_model.py:152-203, encoder_decoder.py:423-680), ACDC-shaped
images + label maps, and injected MaxStyle states.

There is no network on the GPU box, so `bench.py` runs on these ("data": "synthetic").  The functions are value-for-value the generators the parity
tests use (oracle/maxstyle_oracle.py keeps its own copies so that the oracle stays self-contained; tests/test_abi.py::test_synthetic_matches_oracle
checks they agree bit for bit) - the product side never imports the oracle."""
import zlib
from dataclasses import dataclass
from typing import Dict

import numpy as np
import torch


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


@dataclass
class StyleInit:
    """Injected state of one MaxStyle layer (what MaxStyle.__init__ would draw: maxstyle.py:55-117)."""
    perm: torch.Tensor                     # int64 [B]
    lmda: torch.Tensor                     # [B,1,1,1]
    gamma_noise: torch.Tensor              # [B,C,1,1]
    beta_noise: torch.Tensor               # [B,C,1,1]


def random_style_state(batch: int, channels: int, seed: int, dtype=torch.float32) -> StyleInit:
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(batch, generator=g)
    while batch > 1 and torch.equal(perm, torch.arange(batch)):
        perm = torch.randperm(batch, generator=g)
    return StyleInit(perm=perm, lmda=torch.rand(batch, 1, 1, 1, generator=g).to(dtype),
                     gamma_noise=torch.randn(batch, channels, 1, 1, generator=g).to(dtype),
                     beta_noise=torch.randn(batch, channels, 1, 1, generator=g).to(dtype))
