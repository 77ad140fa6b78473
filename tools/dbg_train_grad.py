import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
torch.set_num_threads(min(32, os.cpu_count() or 1))
from test_train_gpu import oracle_pass_grads, make_solver
from parity_util import rel
from oracle import maxstyle_oracle as orc
from oracle import outer_oracle as outer
from maxstyle_amd._lib import lib
dev = torch.device("cuda:0")
o64 = oracle_pass_grads(torch.float64, 16, 256, True)
o32 = oracle_pass_grads(torch.float32, 16, 256, True)
res = {}
for on in (0, 1):
    lib.ms_conv_s2g2_enable(on)
    S, W = make_solver(dev, orc.NetSpec(4, 1, 4))
    S.reset_all_optimizers()
    out = S.standard_training(o32["clean"].to(dev), o32["lab"].to(dev), perturbed_image=o32["image_l"].to(dev), disable_track_bn_stats=False, return_output=True)
    seg, rec = out[0], out[1]
    (seg + rec).backward()
    torch.cuda.synchronize()
    g = {}
    for net in outer.NETS:
        for k, p in S.model[net].named_parameters():
            if o64["grads"][f"{net}/{k}"] is None or outer.is_null_grad_bias(net, k):
                continue
            g[f"{net}/{k}"] = p.grad.detach().cpu().double().clone()
    res[on] = g
    errs = sorted(((rel(v, o64["grads"][k]), k) for k, v in g.items()), reverse=True)
    print("s2g2", on, "top errors vs fp64:", [("%.2e" % e, k.split("/")[-1][-28:]) for e, k in errs[:5]], flush=True)
k = "image_encoder/general_encoder.down4.conv.3.weight"
r64 = o64["grads"][k]
for on in (0, 1):
    d = (res[on][k] - r64).abs()
    sc = float(r64.abs().max())
    print("s2g2", on, k, "rel", rel(res[on][k], r64), "max|d|/max|ref| %.2e" % float(d.max() / sc), "elements with |d| > 1e-3 max|ref|:", int((d > 1e-3 * sc).sum()), "of", d.numel())
d = (res[1][k] - res[0][k]).abs()
print("gen2 - gen1: max %.2e (max|ref| %.2e); per output channel max:" % (float(d.max()), float(r64.abs().max())), [("%.1e" % float(v)) for v in d.flatten(1).max(1).values[:16]])
print("fp32 oracle vs fp64 on it:", rel(o32["grads"][k].double(), r64))
