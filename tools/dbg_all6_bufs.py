"""all-six-layers case: every engine buffer after ONE step, Winograd form against direct form (GPU box) - which tensor of the backward pass the two forms first disagree on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import r3_cases as R
dev = torch.device("cuda:0")
R.ARG_CALLS["all6"] = dict(n_iter=1)
bufs = {}
for wino in ("1", "0"):
    os.environ["MS_LOOP_WINOGRAD"] = wino
    S = R.trained_solver(dev, "trained_fcn16.npz")
    try:
        R.arg_case(dev, "all6", S)
    except Exception as e:
        pass
    eng = next(iter(S._engines.values()))
    torch.cuda.synchronize()
    bufs[wino] = {k: v.detach().float().cpu().clone() for k, v in eng.buf.items() if torch.is_tensor(v) and v.is_floating_point()}
    order = list(eng.buf.keys())
for k in order:
    if k in bufs["1"] and k in bufs["0"] and bufs["1"][k].shape == bufs["0"][k].shape and bufs["1"][k].numel() > 0:
        a, b = bufs["1"][k].double(), bufs["0"][k].double()
        m = float(b.abs().max())
        if m == 0 or not np.isfinite(m):
            continue
        d = float((a - b).abs().max()) / m
        l2 = float((a - b).norm() / max(float(b.norm()), 1e-30))
        print(f"{k:24s} {str(tuple(a.shape)):24s} max|.| {m:.2e}  max-norm diff {d:.2e}  l2 diff {l2:.2e}" + ("   <<<" if d > 1e-4 else ""))
# LeakyReLU masks: elements whose pre-activation sc * u + sh has a different sign in the two runs (the backward multiplies the gradient there by 1 in one run, by 0.2 in the other)
print("== activation masks that differ between the two forms (encoder / segmentor: same parameters in both runs)")
tot = 0
for k in order:
    if not (k.startswith("e.") or k.startswith("s.")) or not (k.endswith(".u1") or k.endswith(".ua") or k.endswith(".u")):
        continue
    ck = k.rsplit(".", 1)[0] + (".bn.coef" if k.endswith(".u") else ".bn1.coef")
    if ck not in bufs["1"] or k not in bufs["0"]:
        continue
    pre = {}
    for w in ("1", "0"):
        c = bufs[w][ck].double()
        pre[w] = c[:, 0].view(1, -1, 1, 1) * bufs[w][k].double() + c[:, 1].view(1, -1, 1, 1)
    diff = (pre["1"] > 0) != (pre["0"] > 0)
    n = int(diff.sum()); tot += n
    line = f"   {k:12s} {pre['1'].numel():8d} elements, masks differ at {n}"
    if n:
        idx = diff.nonzero()[:4].tolist()
        line += "  " + "; ".join(f"{tuple(i)}: pre {float(pre['1'][tuple(i)]):+.2e} (Winograd) {float(pre['0'][tuple(i)]):+.2e} (direct)" for i in idx)
    print(line)
print("   total", tot)
