"""Where does the HOST block inside a generate_max_style_image call when calls are issued back to back (deferred error protocol)?  Wall-clock of the entry / exit of
the call's main pieces, relative to the call's entry, averaged over the last calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maxstyle_amd as M
from maxstyle_amd import synthetic as syn, engine as E, maxstyle as MS
dev = torch.device("cuda:0")
spec = syn.NetSpec(4, 1, 4)
S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True)
W = syn.procedural_weights(spec, 0)
for name, mod in S.model.items():
    mod.load_state_dict(W[name]); mod.train()
S.loop_error_check = "deferred"
img, lab = syn.synthetic_batch(16, 256, 1, 4, 1234)
img, lab = img.to(dev), lab.to(dev)
z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
z_i = z_i.detach()
marks = []
t_call = [0.0]


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((label, t0 - t_call[0], time.perf_counter() - t_call[0]))
        return r
    setattr(obj, name, g)


wrap(MS.MaxStyle, "__init__", "MaxStyle()")
wrap(E.InnerLoopEngine, "set_style_states", "set_style_states")
wrap(E.InnerLoopEngine, "run", "run")
wrap(E.InnerLoopEngine, "check_errors", "check_errors")
wrap(E.InnerLoopEngine, "restore_config", "restore_config")
wrap(E.InnerLoopEngine, "stash_config", "stash_config")
wrap(torch, "_foreach_copy_", "foreach_copy")
res = []
for it in range(14):
    marks.clear()
    t_call[0] = time.perf_counter()
    S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, p=1.5, n_iter=5, lr=0.1, reference_image=img, reference_segmentation=lab)
    res.append((time.perf_counter() - t_call[0], list(marks)))
torch.cuda.synchronize()
for tot, mk in res[-3:]:
    print(f"call: host {tot * 1e3:.3f} ms")
    for lab_, a, b in mk:
        print(f"    {lab_:18s} {a * 1e3:8.3f} -> {b * 1e3:8.3f} ms   ({(b - a) * 1e3:.3f})")
