"""encode -> generate -> encode ...: the encoder engine's buffers after each encode, first differing buffer (scratch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from maxstyle_amd.options import set_library_option, engine_defaults
from maxstyle_amd import synthetic as syn
import r4_cases as R4
from collections import Counter
dev = torch.device("cuda:0")
set_library_option("conv.wino_flat", int(sys.argv[1]))
n = int(sys.argv[2])
gen = len(sys.argv) <= 3 or sys.argv[3] != "nogen"
spec, size = syn.NetSpec(1, 3, 2), 320
runs = []
with engine_defaults(winograd=True):
    S = R4.trained_solver64(dev)
    enc = S.model['image_encoder']
    B, layers = 16, [3, 4, 5]
    img, lab = syn.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
    styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = R4._inject_all(styles, dev)
    for i in range(n):
        z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
        torch.cuda.synchronize()
        eng = next(iter(enc._engines.values()))
        r = {k: v.detach().clone() for k, v in eng.buf.items() if torch.is_tensor(v) and v.is_floating_point() and not k.endswith(".stats")}
        for k, v in eng.buf.items():
            if torch.is_tensor(v) and k.endswith(".stats"):
                r[k] = v.detach()[1:].clone()      # (without the header: its launch epoch counts up)
        r["~z_i"] = z_i.clone()
        runs.append(r)
        if gen:
            S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=1, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
            torch.cuda.synchronize()
sig = [float(r["~z_i"].double().sum()) for r in runs]
good = runs[sig.index(Counter(sig).most_common(1)[0][0])]
print("outcomes (sum of z_i):", Counter(sig))
order = list(good.keys())
for i, r in enumerate(runs):
    bad = [(k, float((r[k] - good[k]).abs().max()) / max(float(good[k].abs().max()), 1e-30), int((r[k] != good[k]).sum()), r[k].numel()) for k in order if k in r and r[k].shape == good[k].shape and not torch.equal(r[k], good[k])]
    if bad:
        print(f"call {i}: {len(bad)} of {len(good)} buffers differ", flush=True)
        for x_ in bad[:8]:
            print("     %-28s rel %.2e  elements %d / %d" % x_, flush=True)
        k = bad[0][0]
        d = (r[k] != good[k]).nonzero()
        print("      first differing buffer", k, tuple(good[k].shape), "first/last index", d[0].tolist(), d[-1].tolist())
        if good[k].dim() == 2:
            print("      channels (table rows):", sorted(set(d[:, 0].div(2048, rounding_mode="floor").tolist()))[:60], "slots:", sorted(set((d[:, 0] % 2048).tolist()))[:40], "fields", sorted(set(d[:, 1].tolist())))
            for q in d[:5].tolist():
                print("        ", q, r[k][q[0]].tolist(), good[k][q[0]].tolist())
        elif good[k].dim() == 4:
            print("      images:", sorted(set(d[:, 0].tolist())), "channels:", len(set(d[:, 1].tolist())), sorted(set(d[:, 1].tolist()))[:40], "rows:", sorted(set(d[:, 2].tolist())), "cols:", sorted(set(d[:, 3].tolist())))
            for q in d[:5].tolist():
                print("        ", q, r[k][tuple(q)].item(), good[k][tuple(q)].item())
print("order:", order)
