"""Host side of one generate_max_style_image call (C2 workload): cProfile of 50 calls with the deferred error protocol, n_iter = 5 and n_iter = 0 (GPU box)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maxstyle_amd as M
from maxstyle_amd import synthetic as syn
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
call = lambda k: S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, p=1.5, n_iter=k, lr=0.1, reference_image=img, reference_segmentation=lab, fix_seed=7)
for k in (5, 0):
    for _ in range(5):
        call(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        call(k)
    t_host = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 50
    print(f"n_iter={k}: host returns after {t_host * 1e3:.3f} ms per call; with the GPU drained {t_all * 1e3:.3f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        call(k)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(l[:160] for l in s.getvalue().splitlines()[:60]))
