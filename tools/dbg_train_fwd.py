import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from test_train_gpu import make_solver
from oracle import maxstyle_oracle as orc
from oracle import outer_oracle as outer
from maxstyle_amd._lib import lib
dev = torch.device("cuda:0")
clean, lab = orc.synthetic_batch(16, 256, 1, 4, 1234)
g = torch.Generator().manual_seed(100)
image_l = outer.noisy_input(clean, 0.05 * torch.randn(clean.shape, generator=g))
bufs = {}
grads = {}
for on in (0, 1):
    lib.ms_conv_s2g2_enable(on)
    S, W = make_solver(dev, orc.NetSpec(4, 1, 4))
    S.reset_all_optimizers()
    out = S.standard_training(clean.to(dev), lab.to(dev), perturbed_image=image_l.to(dev), disable_track_bn_stats=False, return_output=True)
    eng = [e for pool in S._train_engines.values() for e in pool][0]
    torch.cuda.synchronize()
    bufs[on] = {k: v.detach().clone() for k, v in eng.buf.items() if torch.is_tensor(v) and v.dtype == torch.float32 and v.numel() > 1000}
    (out[0] + out[1]).backward()
    torch.cuda.synchronize()
    bufs[(on, "b")] = {k: v.detach().clone() for k, v in eng.buf.items() if torch.is_tensor(v) and v.dtype == torch.float32 and v.numel() > 1000}
for tag in (0, "b"):
    a, b = (bufs[0], bufs[1]) if tag == 0 else (bufs[(0, "b")], bufs[(1, "b")])
    rows = []
    for k in a:
        if k in b and a[k].shape == b[k].shape:
            d = (a[k] - b[k]).abs()
            sc = float(a[k].abs().max()) + 1e-30
            rows.append((float(d.max()) / sc, k, int((d > 1e-3 * sc).sum()), a[k].numel()))
    rows.sort(reverse=True)
    print("== after", "forward" if tag == 0 else "backward", ": largest relative differences between the two runs (max|d|/max|a|, buffer, elements > 1e-3, size)")
    for r in rows[:14]:
        print("   %.2e %-22s %8d / %d" % r)
