import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
torch.set_num_threads(min(32, os.cpu_count() or 1))
from test_train_gpu import oracle_pass_grads, make_solver
from parity_util import rel
from oracle import maxstyle_oracle as orc
from maxstyle_amd._lib import lib
dev = torch.device("cuda:0")
o64 = oracle_pass_grads(torch.float64, 16, 256, True)
o32 = oracle_pass_grads(torch.float32, 16, 256, True)
print("fp32 oracle vs fp64: z_i %.2e recon %.2e logits %.2e" % (rel(o32["z_i"].double(), o64["z_i"]), rel(o32["recon"].double(), o64["recon"]), rel(o32["logits"].double(), o64["logits"])))
for on in (0, 1):
    lib.ms_conv_s2g2_enable(on)
    S, W = make_solver(dev, orc.NetSpec(4, 1, 4))
    S.reset_all_optimizers()
    out = S.standard_training(o32["clean"].to(dev), o32["lab"].to(dev), perturbed_image=o32["image_l"].to(dev), disable_track_bn_stats=False, return_output=True)
    seg, rec, _, _, recon, y0, _ = out
    print("s2g2", on, "vs fp64: z_i %.2e recon %.2e logits %.2e  seg %.3e rec %.3e" % (rel(S.z_i.cpu().double(), o64["z_i"]), rel(recon.cpu().double(), o64["recon"]), rel(y0.cpu().double(), o64["logits"]),
          abs(float(seg) - o64["seg"]) / abs(o64["seg"]), abs(float(rec) - o64["rec"]) / abs(o64["rec"])))
