"""CPU oracle for one OUTER training iteration around the MaxStyle inner loop.

TEST INFRASTRUCTURE ONLY (same rules as maxstyle_oracle.py: imported by tests/, smoke() and bench.py's cpu_baseline leg only).

Plain-PyTorch restatement (autograd over the functional forward of maxstyle_oracle.py) of, all paths under /root/reference/src:
  noisy_input            train_adv_supervised_segmentation_triplet.py:177-181   image_l = clamp(clean + 0.05*randn, clean.min(), clean.max())
  training_pass          models/advanced_triplet_recon_segmentation_model.py:731-786 (standard_training, 'no_STN' networks):
                         seg loss = cross_entropy_2D(seg_decoder(z_s)), recon loss = 0.5*MSE(image_decoder(z_i), clean)
  hard pass              :843-889 (hard_example_traininng): rescale_intensity(stylised, 0, 1) -> standard_training with
                         disable_track_bn_stats=True: batch statistics, no running update, BatchNorm affine frozen (model_util.py:468-510)
  rescale_intensity      common_utils/basic_operations.py:257-281
  bn running update      torch.nn.BatchNorm2d train mode: running = 0.9*running + 0.1*batch (unbiased variance), num_batches_tracked += 1
  adamw_step             torch.optim.AdamW defaults (lr, betas (0.9,0.999), eps 1e-8, weight_decay 1e-2); one optimiser per sub-net (:1055-1063)
  train_iteration        train_adv_supervised_segmentation_triplet.py:163-199, 251-287, 532-535:
                         loss = standard (seg + recon) + max-style (seg + recon on the stylised image); backward; AdamW step x3
Pinned by tests/golden/outer_*.npz (reference run in the build container by tests/golden/make_golden_outer.py).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from . import maxstyle_oracle as orc

NETS = ("image_encoder", "segmentation_decoder", "image_decoder")


def rescale_intensity(data, new_min=0.0, new_max=1.0, eps=1e-20):
    bs, c = data.shape[0], data.shape[1]
    flat = data.reshape(bs * c, -1)
    old_max = flat.max(dim=1, keepdim=True).values
    old_min = flat.min(dim=1, keepdim=True).values
    out = (flat - old_min) / (old_max - old_min + eps) * (new_max - new_min) + new_min
    return out.view_as(data)


def noisy_input(clean, noise):
    return torch.clamp(clean + noise, clean.min(), clean.max())


def is_bn_affine(sd, name):
    return (name.endswith(".weight") or name.endswith(".bias")) and (name.rsplit(".", 1)[0] + ".running_mean") in sd


def is_null_grad_bias(net, name):
    """Bias of a convolution that feeds a batch-statistics BatchNorm: the loss does not depend on it (the batch mean removes it), its exact
    gradient is 0 and what the reference computes is summation round-off (1e-15 in fp64, 1e-7 in fp32) that Adam then normalises to +-lr."""
    if not name.endswith(".bias"):
        return False
    stem = name[:-5]
    return stem.endswith(".conv.0") or stem.endswith(".conv.3") or stem.endswith("inc.0") or stem.endswith("inc.3") or \
        (net == "image_encoder" and stem.endswith("final_conv.0"))


def param_names(sd):
    """Learnable tensors in nn.Module.parameters() order == state_dict order without the BatchNorm buffers."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def _running_update(sd, name, u, momentum=0.1):
    with torch.no_grad():
        n = u.numel() // u.shape[1]
        mean = u.mean(dim=(0, 2, 3))
        var = u.var(dim=(0, 2, 3), unbiased=True) if n > 1 else torch.zeros_like(mean)
        sd[name + ".running_mean"].mul_(1 - momentum).add_(mean, alpha=momentum)
        sd[name + ".running_var"].mul_(1 - momentum).add_(var, alpha=momentum)
        sd[name + ".num_batches_tracked"] += 1


def training_pass(weights, perturbed_image, clean_image, labels, track_bn: bool, mix=None):
    """standard_training for a 'no_STN' network. track_bn=True: the clean pass (running statistics updated, BatchNorm affine learns);
    track_bn=False: the pass inside _disable_tracking_bn_stats (BatchNorm affine is a constant of the graph)."""
    w = weights
    if not track_bn:
        w = {net: {k: (v.detach() if is_bn_affine(sd, k) else v) for k, v in sd.items()} for net, sd in weights.items()}
    prev = orc.BN_OBSERVER
    orc.BN_OBSERVER = (lambda sd, name, u: _running_update(sd, name, u)) if track_bn else None
    try:
        z_i, z_s = orc.encoder_forward(w["image_encoder"], perturbed_image, "batch", mix=mix)
        logits = orc.decoder_forward(w["segmentation_decoder"], z_s, "NN", None, "batch")
        recon = orc.decoder_forward(w["image_decoder"], z_i, "Conv2", "sigmoid", "batch")
    finally:
        orc.BN_OBSERVER = prev
    seg_loss = orc.cross_entropy_2d(logits, labels)
    recon_loss = 0.5 * torch.mean((recon - clean_image.detach()) ** 2)
    return seg_loss, recon_loss, z_i, z_s, recon, logits


def adamw_step(p, g, m, v, t, lr, wd=1e-2, b1=0.9, b2=0.999, eps=1e-8, decoupled=True):
    """torch.optim.AdamW (decoupled=True) / torch.optim.Adam with weight_decay=0 (decoupled=False, wd ignored). In place."""
    if decoupled:
        p.mul_(1 - lr * wd)
    orc.adam_step(p, g, m, v, t, lr, b1, b2, eps)


def new_optimizer_state(weights):
    return {net: {k: (torch.zeros_like(weights[net][k]), torch.zeros_like(weights[net][k])) for k in param_names(weights[net])} for net in NETS}, {"t": 0}


def train_iteration(weights, opt_state, clean, labels, noise, styles: Dict[int, "orc.StyleState"], layers: Sequence[int],
                    n_iter=5, lr_inner=0.1, lr_outer=1e-4, optimizer="AdamW"):
    """One iteration of the MaxStyle training loop on `weights` (dict net -> state_dict of plain tensors, updated IN PLACE).
    Returns a dict with the four losses, the stylised image and the parameter gradients."""
    for net in NETS:
        for k in param_names(weights[net]):
            weights[net][k].requires_grad_(True)
            weights[net][k].grad = None
    image_l = noisy_input(clean, noise)
    seg0, rec0, z_i, z_s, recon0, logits0 = training_pass(weights, image_l, clean, labels, track_bn=True)
    frozen = {net: {k: v.detach() for k, v in sd.items()} for net, sd in weights.items()}
    stylised = orc.generate_max_style_image(frozen, z_i.detach(), styles, layers, labels, n_iter=n_iter, lr=lr_inner)
    hard_in = rescale_intensity(stylised.detach().clone(), 0.0, 1.0)
    seg1, rec1, _, _, recon1, logits1 = training_pass(weights, hard_in, clean, labels, track_bn=False)
    loss = (seg0 + rec0) + (rec1 + seg1)
    names = [(net, k) for net in NETS for k in param_names(weights[net])]
    grads = torch.autograd.grad(loss, [weights[n][k] for n, k in names], allow_unused=True)
    state, counter = opt_state
    counter["t"] += 1
    out_g = {}
    with torch.no_grad():
        for (net, k), g in zip(names, grads):
            p = weights[net][k]
            p.requires_grad_(False)
            if g is None:
                continue
            out_g[f"{net}/{k}"] = g.detach().clone()
            m, v = state[net][k]
            adamw_step(p, g, m, v, counter["t"], lr_outer, decoupled=(optimizer == "AdamW"))
    return {"seg_loss": float(seg0.detach()), "recon_loss": float(rec0.detach()), "hard_seg_loss": float(seg1.detach()), "hard_recon_loss": float(rec1.detach()),
            "image_l": image_l.detach(), "z_i": z_i.detach(), "stylised": stylised.detach(), "hard_input": hard_in.detach(),
            "recon": recon0.detach(), "grads": out_g}
