"""Where one trainer iteration goes (GPU box): wall time of each phase of bench.py's outer_iteration with a device synchronisation between the phases, and the same
iteration unsynchronised.   python tools/train_timeline.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maxstyle_amd as M
from maxstyle_amd import synthetic as syn


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
    S.loop_error_check = "deferred"
    clean, lab = syn.synthetic_batch(16, 256, 1, 4, 1234)
    clean, lab = clean.to(dev), lab.to(dev)
    cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 5, "mix_learnable": True, "noise_learnable": True,
           "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}
    acc = {}

    def phase(name, fn, sync):
        if sync:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        th = time.perf_counter()
        if sync:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        a = acc.setdefault(name, [0.0, 0.0])
        a[0] += t1 - t0; a[1] += th - t0
        return out

    def iteration(sync):
        S.train()
        S.reset_all_optimizers()
        image_l = phase("noise", lambda: torch.clamp(clean + 0.05 * torch.randn_like(clean), clean.min(), clean.max()), sync)
        o = phase("standard_training", lambda: S.standard_training(clean, lab, perturbed_image=image_l, return_output=True), sync)
        seg0, rec0, gt0, sh0 = o[0], o[1], o[2], o[3]
        S.reset_all_optimizers()
        sty = phase("generate_max_style_image", lambda: S.generate_max_style_image_from_config(S.z_i, cfg, clean, lab, p=1.5).detach().clone(), sync)
        seg1, rec1, sh1, sh2 = phase("hard_example_training", lambda: S.hard_example_traininng(perturbed_image=sty, perturbed_seg=None, clean_image_l=clean, label_l=lab), sync)
        loss = (seg0 + rec0 + sh0 + gt0) + (rec1 + seg1 + sh1 + sh2)
        S.reset_all_optimizers()
        phase("backward", lambda: loss.backward(), sync)
        phase("optimize_all_params", lambda: S.optimize_all_params(), sync)
        return loss

    for _ in range(3):
        iteration(False)
    acc.clear()
    for _ in range(iters):
        iteration(True)
    print(f"# per iteration, {iters} iterations, device synchronised around every phase: wall ms (host-side ms before the call returned)")
    tot = 0.0
    for k, (w, h) in acc.items():
        print(f"{k:28s} {w / iters * 1e3:8.3f}  ({h / iters * 1e3:7.3f})")
        tot += w
    print(f"{'sum':28s} {tot / iters * 1e3:8.3f}")
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration(False)
    torch.cuda.synchronize()
    print(f"unsynchronised iteration: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms; host side per phase (ms):")
    for k, (w, h) in acc.items():
        print(f"  {k:28s} {h / iters * 1e3:8.3f}")


if __name__ == "__main__":
    main()
