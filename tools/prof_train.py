"""One TrainEngine forward + backward at config 2, repeated (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maxstyle_amd as M
from maxstyle_amd import synthetic as syn

dev = torch.device("cuda:0")
S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
B, size = 16, 256
clean, lab = syn.synthetic_batch(B, size, 1, 4, 1234)
clean, lab = clean.to(dev), lab.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for it in range(n):
    S.reset_all_optimizers()
    seg, rec, _, _ = S.standard_training(clean, lab, perturbed_image=clean)
    (seg + rec).backward()
torch.cuda.synchronize()
print("done")
