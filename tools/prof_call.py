# (round 3 diagnostic; see profiles/r03_experiments.txt 17-18 and DESIGN.md section 4)
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import maxstyle_amd as M
from maxstyle_amd import synthetic as syn
dev = torch.device("cuda:0")
S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True)
for m in S.model.values(): m.train()
B = 16
img, lab = syn.synthetic_batch(B, 256, 1, 4, seed=1234)
img, lab = img.to(dev), lab.to(dev)
z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
z_i = z_i.detach()
S.loop_error_check = os.environ.get("MS_ERROR_CHECK", "sync")
def call(K):
    return S.generate_max_style_image(z_i, [3, 4, 5], [128, 64, 32, 16, 16, 1], p=1.5, n_iter=K, lr=0.1, reference_image=img, reference_segmentation=lab)
for K in (5, 0):
    for _ in range(5): call(K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): call(K)
    torch.cuda.synchronize(); print("K", K, "ms per call", (time.perf_counter() - t0) / 30 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(30): call(5)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
