import os, sys
sys.path.insert(0, os.getcwd())
import torch
import maxstyle_amd as M
from oracle import maxstyle_oracle as orc
dev = torch.device("cuda:0")
S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
B, size = 16, 256
clean, lab = orc.synthetic_batch(B, size, 1, 4, 1234)
clean, lab = clean.to(dev), lab.to(dev)
cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 5, "mix_learnable": True, "noise_learnable": True,
       "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}
torch.manual_seed(0)
for it in range(60):
    S.train(); S.reset_all_optimizers()
    image_l = torch.clamp(clean + 0.05 * torch.randn_like(clean), clean.min(), clean.max())
    seg0, rec0, gt0, sh0, recon0, p0, _ = S.standard_training(clean, lab, perturbed_image=image_l, return_output=True)
    kw = dict(image_code=S.z_i, channel_num=[128, 64, 32, 16, 16, 1], p=1.5, decoder_layers_indexes=[3, 4, 5], lr=0.1, reference_image=clean, reference_segmentation=lab)
    sty = S.generate_max_style_image(n_iter=5, fix_seed=1000 + it, **kw).detach().clone()
    if not bool(torch.isfinite(sty).all()):
        eng = list(S._engines.values())[0]
        def report(tag):
            torch.cuda.synchronize()
            bad = [k for k, v in eng.buf.items() if v.is_floating_point() and not bool(torch.isfinite(v).all())]
            print("  ", tag, "non-finite buffers:", bad[:12], "flat_p", bool(torch.isfinite(eng.flat_p).all()), "flat_g", bool(torch.isfinite(eng.flat_g).all()),
                  "losses", eng.loss_buf[:3].tolist(), "ws err", int(eng.buf["style.ws"].view(torch.int32)[1]))
        snaps = {}
        for tag, graph in (("eager", False), ("graph", True), ("eager_b", False), ("graph_b", True)):
            S.generate_max_style_image(n_iter=1, fix_seed=1000 + it, use_graph=graph, **kw)
            torch.cuda.synchronize()
            snaps[tag] = {k: v.clone() for k, v in eng.buf.items() if v.is_floating_point()}
            snaps[tag]["flat_p"] = eng.flat_p.clone(); snaps[tag]["flat_g"] = eng.flat_g.clone()
        for other in ("graph", "eager_b", "graph_b"):
            diffs = []
            for k, v in snaps["eager"].items():
                w = snaps[other].get(k)
                if w is None or w.shape != v.shape: continue
                d = float((v - w).abs().max())
                if d != 0.0 or not bool(torch.isfinite(w).all()): diffs.append((k, d))
            print("  eager vs", other, "differing buffers:", len(diffs), diffs[:40])
    seg1, rec1, sh1, sh2 = S.hard_example_traininng(perturbed_image=sty, perturbed_seg=None, clean_image_l=clean, label_l=lab)
    loss = (seg0 + rec0) + (rec1 + seg1)
    S.reset_all_optimizers()
    loss.backward()
    b = S._bank
    gf = bool(torch.isfinite(b.flat_g).all())
    if it % 5 == 0 or not gf: print(it, [round(float(v.detach()), 5) for v in (seg0, rec0, seg1, rec1)], "grad finite", gf, "gnorm", float(b.flat_g.norm()))
    if not gf:
        for (net, name), (o, n, shp) in b.index.items():
            if not bool(torch.isfinite(b.flat_g[o:o+n]).all()): print("   non-finite grad:", net, name)
        break
    S.optimize_all_params()
