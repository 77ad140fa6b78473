# (round 3 diagnostic; see profiles/r03_experiments.txt 17-18 and DESIGN.md section 4)
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_train_gpu as T
from parity_util import rel
from oracle import maxstyle_oracle as orc, outer_oracle as outer
torch.set_num_threads(min(32, os.cpu_count() or 1))
dev = torch.device("cuda:0")
which = sys.argv[1]
cache = "/tmp/o_%s.pt" % which
dt = torch.float64 if which == "f64" else torch.float32
o = T.oracle_pass_grads(dt, 16, 256, True)
S, W = T.make_solver(dev, orc.NetSpec(4, 1, 4))
S.reset_all_optimizers()
out = S.standard_training(o["clean"].float().to(dev), o["lab"].to(dev), perturbed_image=o["image_l"].float().to(dev), disable_track_bn_stats=False, return_output=True)
seg, rec = out[0], out[1]
(seg + rec).backward()
errs = []
for net in outer.NETS:
    for k, p in S.model[net].named_parameters():
        ref = o["grads"][f"{net}/{k}"]
        if ref is None or outer.is_null_grad_bias(net, k): continue
        gd = p.grad.detach().cpu().double(); rd = ref.double()
        errs.append((rel(p.grad, ref.float()), f"{net}/{k}", float((gd - rd).norm() / rd.norm())))
errs.sort(reverse=True)
print(which, os.environ.get("MS_LIB", "default")[-30:])
print('  worst L2 over all tensors: %.3e' % max(e[2] for e in errs))
for e in errs[:4]: print("  max-norm %.3e  %s  L2 %.3e" % e)
