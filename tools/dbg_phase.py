import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import maxstyle_amd as M
from oracle import maxstyle_oracle as orc
dev = torch.device("cuda:0")
S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
B, size = 16, 256
clean, lab = orc.synthetic_batch(B, size, 1, 4, 1234)
clean, lab = clean.to(dev), lab.to(dev)
def T(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
S.train(); S.reset_all_optimizers()
out = {}
def std():
    return S.standard_training(clean, lab, perturbed_image=clean, return_output=True)
out["standard_fwd (no repack)"] = T(std)
def repack():
    S._weights_epoch += 1
    S._loop_engine(B, size, size, dev)
out["repack"] = T(repack)
def fwd_bwd():
    S.reset_all_optimizers()
    r = std()
    (r[0] + r[1]).backward()
out["fwd+bwd"] = T(fwd_bwd)
eng = S._train_engines[(B, size, size, str(dev))][0]
def fwd_only_engine():
    eng.forward_pass(clean, lab, clean, False, None)
out["engine.forward_pass"] = T(fwd_only_engine)
def bwd_only_engine():
    eng.backward_pass(clean, lab, clean, 1.0, 1.0)
out["engine.backward_pass"] = T(bwd_only_engine)
# GPU time of the same: capture into graphs
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.forward_pass(clean, lab, clean, False, None)
out["forward_pass graph replay"] = T(g.replay)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    eng.backward_pass(clean, lab, clean, 1.0, 1.0)
out["backward_pass graph replay"] = T(g2.replay)
for k, v in out.items(): print(f"{k:32s} {v:8.3f} ms")
