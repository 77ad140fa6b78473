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
for seed in (100, 101, 102, 103, 104, 105):
    o64 = oracle_pass_grads(torch.float64, 16, 256, True, seed_noise=seed)
    o32 = oracle_pass_grads(torch.float32, 16, 256, True, seed_noise=seed)
    ks = [k for k, g in o32["grads"].items() if g is not None and not outer.is_null_grad_bias(*k.split("/", 1))]
    line = "seed %d  oracle32 worst max-norm %.1e L2 %.1e" % (seed, max(rel(o32["grads"][k].double(), o64["grads"][k]) for k in ks),
                                                             max(float((o32["grads"][k].double() - o64["grads"][k]).norm() / o64["grads"][k].norm()) for k in ks))
    for on in (0, 1):
        lib.ms_set_option(b"conv.s2g2", int(on))
        S, W = make_solver(dev, orc.NetSpec(4, 1, 4))
        S.reset_all_optimizers()
        out = S.standard_training(o32["clean"].to(dev), o32["lab"].to(dev), perturbed_image=o32["image_l"].to(dev), disable_track_bn_stats=False, return_output=True)
        (out[0] + out[1]).backward()
        worst = ("", 0.0)
        worst2 = 0.0
        for net in outer.NETS:
            for k, p in S.model[net].named_parameters():
                ref = o64["grads"][f"{net}/{k}"]
                if ref is None or outer.is_null_grad_bias(net, k):
                    continue
                gd = p.grad.cpu().double()
                e = rel(gd, ref)
                worst2 = max(worst2, float((gd - ref).norm() / ref.norm()))
                if e > worst[1]:
                    worst = (k[-26:], e)
        line += "  | s2g2 %d worst max-norm %.1e %s, worst L2 %.1e" % (on, worst[1], worst[0], worst2)
    print(line, flush=True)
