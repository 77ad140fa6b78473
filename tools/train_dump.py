# (round 3 diagnostic; see profiles/r03_experiments.txt 17-18 and DESIGN.md section 4)
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_train_gpu as T
from oracle import maxstyle_oracle as orc, outer_oracle as outer
dev = torch.device("cuda:0")
clean, lab = orc.synthetic_batch(16, 256, 1, 4, 1234)
g = torch.Generator().manual_seed(100)
noise = 0.05 * torch.randn(clean.shape, generator=g)
image_l = outer.noisy_input(clean, noise)
S, W = T.make_solver(dev, orc.NetSpec(4, 1, 4))
S.reset_all_optimizers()
out = S.standard_training(clean.to(dev), lab.to(dev), perturbed_image=image_l.to(dev), disable_track_bn_stats=False, return_output=True)
(out[0] + out[1]).backward()
d = {f"{n}/{k}": p.grad.detach().cpu().clone() for n in outer.NETS for k, p in S.model[n].named_parameters() if p.grad is not None}
eng = next(iter(S._train_engines.values()))
eng = eng[0] if isinstance(eng, (list, tuple)) else eng
for k, v in eng.buf.items():
    if torch.is_tensor(v) and v.dtype == torch.float32 and v.numel() > 16 and (k.startswith("e.d4") or k.startswith("e.fc") or k.startswith("e.cd") or k.startswith("e.d3")):
        d["buf:" + k] = v.detach().cpu().clone()
torch.save(d, sys.argv[1])
