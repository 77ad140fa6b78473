"""Whole training iteration at config 2 on one MI355X: standard pass -> MaxStyle inner loop (K=5) -> hard-example pass -> backward -> AdamW
(train_adv_supervised_segmentation_triplet.py:163-199, 251-287, 532-535).  Prints ms per phase."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    import maxstyle_amd as M
    from maxstyle_amd import synthetic as syn
    dev = torch.device("cuda:0")
    spec = syn.NetSpec(4, 1, 4)
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
    clean, lab = syn.synthetic_batch(args.batch, args.size, 1, 4, 1234)
    clean, lab = clean.to(dev), lab.to(dev)
    cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 5, "mix_learnable": True, "noise_learnable": True,
           "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}
    phases = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        phases[name] = phases.get(name, 0.0) + (time.perf_counter() - t0)

    def iteration(record):
        S.train()
        S.reset_all_optimizers()
        t0 = time.perf_counter()
        noise = 0.05 * torch.randn_like(clean)
        image_l = torch.clamp(clean + noise, clean.min(), clean.max())
        seg0, rec0, gt0, sh0, recon0, p0, _ = S.standard_training(clean, lab, perturbed_image=image_l, return_output=True)
        if record: lap("standard_fwd", t0)
        t0 = time.perf_counter()
        S.reset_all_optimizers()
        sty = S.generate_max_style_image_from_config(S.z_i, cfg, clean, lab, p=1.5).detach().clone()
        if record: lap("inner_loop_K5", t0)
        t0 = time.perf_counter()
        seg1, rec1, sh1, sh2 = S.hard_example_traininng(perturbed_image=sty, perturbed_seg=None, clean_image_l=clean, label_l=lab)
        if record: lap("hard_fwd", t0)
        t0 = time.perf_counter()
        loss = (seg0 + rec0 + sh0 + gt0) + (rec1 + seg1 + sh1 + sh2)
        S.reset_all_optimizers()
        loss.backward()
        if record: lap("backward_x2", t0)
        t0 = time.perf_counter()
        S.optimize_all_params()
        if record: lap("adamw", t0)
        return loss

    for _ in range(3):
        iteration(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        loss = iteration(False)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / args.iters
    for _ in range(args.iters):
        iteration(True)
    out = {"ms_per_iteration": round(total * 1e3, 2), "iterations_per_s": round(1 / total, 2), "loss": float(loss),
           "phases_ms_synchronised": {k: round(v / args.iters * 1e3, 2) for k, v in phases.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
