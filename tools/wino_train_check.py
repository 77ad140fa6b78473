"""Weight gradients of one full-size training pass (16x1x256x256): GPU (direct / Winograd form) and the fp32 CPU oracle, each against the fp64 CPU oracle.
python tools/wino_train_check.py   (runs itself twice as a child with MS_OPTIONS=engine.train_winograd=0 / 1)"""
import os, sys, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from test_train_gpu import make_solver
    from oracle import maxstyle_oracle as orc, outer_oracle as outer
    o = pickle.load(open("/tmp/wtc_in.pkl", "rb"))
    dev = torch.device("cuda:0")
    S, W = make_solver(dev, orc.NetSpec(4, 1, 4))
    S.reset_all_optimizers()
    out = S.standard_training(o["clean"].to(dev), o["lab"].to(dev), perturbed_image=o["image_l"].to(dev), disable_track_bn_stats=False, return_output=True)
    seg, rec = out[0], out[1]
    (seg + rec).backward()
    g = {f"{n}/{k}": p.grad.detach().cpu() for n in outer.NETS for k, p in S.model[n].named_parameters()}
    pickle.dump(g, open(sys.argv[2], "wb"))
    sys.exit(0)
from test_train_gpu import oracle_pass_grads
from parity_util import rel
from oracle import outer_oracle as outer
torch.set_num_threads(min(32, os.cpu_count() or 1))
o32 = oracle_pass_grads(torch.float32, 16, 256, True)
o64 = oracle_pass_grads(torch.float64, 16, 256, True)
pickle.dump(dict(clean=o32["clean"], lab=o32["lab"], image_l=o32["image_l"]), open("/tmp/wtc_in.pkl", "wb"))
res = {}
for w in ("0", "1"):
    env = dict(os.environ, MS_OPTIONS="engine.train_winograd=" + w)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", f"/tmp/wtc_{w}.pkl"], env=env)
    res[w] = pickle.load(open(f"/tmp/wtc_{w}.pkl", "rb"))
def worst(get):
    ws = []
    for key, ref in o64["grads"].items():
        n, k = key.split("/", 1)
        if ref is None or outer.is_null_grad_bias(n, k):
            continue
        ws.append((rel(get(key), ref), key))
    ws.sort(reverse=True)
    return ws[:3], sum(e for e, _ in ws) / len(ws)
print("fp32 CPU oracle vs fp64:", worst(lambda k: o32["grads"][k]))
print("GPU direct      vs fp64:", worst(lambda k: res["0"][k]))
print("GPU Winograd    vs fp64:", worst(lambda k: res["1"][k]))
