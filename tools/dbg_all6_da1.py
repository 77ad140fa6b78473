"""all-six-layers case: the inputs of the first launch the two conv forms disagree on (e.d1.da1: data-gradient 32 -> 32 @32x32 with the BatchNorm-backward prologue and the
activation-backward epilogue), captured from the engine, then run alone in both forms against fp64 math (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
import r3_cases as R
from maxstyle_amd import engine as E, ops
dev = torch.device("cuda:0")
R.ARG_CALLS["all6"] = dict(n_iter=1)
NAME = sys.argv[1] if len(sys.argv) > 1 else "e.d1.da1"
cap = {}
orig = E.Engine.conv_actbwd if hasattr(E, "Engine") and hasattr(E.Engine, "conv_actbwd") else E.InnerLoopEngine.conv_actbwd
owner = E.Engine if hasattr(E, "Engine") and hasattr(E.Engine, "conv_actbwd") else E.InnerLoopEngine
def wrapped(self, name, bw_name, g, cw, bnbwd, u, coef, slope):
    if name == NAME and "g" not in cap:
        torch.cuda.synchronize()
        cap.update(g=g.clone(), bcoef=(bnbwd[0].clone() if torch.is_tensor(bnbwd[0]) else bnbwd[0]), u2=bnbwd[1].clone(), u=u.clone(), coef=coef.clone(), cw=cw, slope=slope)
    out = orig(self, name, bw_name, g, cw, bnbwd, u, coef, slope)
    if name == NAME and "out" not in cap:
        torch.cuda.synchronize(); cap["out"] = out[0].clone()
    return out
owner.conv_actbwd = wrapped
S = R.trained_solver(dev, "trained_fcn16.npz")
try:
    R.arg_case(dev, "all6", S)
except Exception as e:
    pass
print("captured:", {k: (tuple(v.shape) if torch.is_tensor(v) else type(v).__name__) for k, v in cap.items()})
cw = cap["cw"]; g, u2, u, coef, bc = cap["g"], cap["u2"], cap["u"], cap["coef"], cap["bcoef"]
print("cw: cin", cw.cin, "cout", cw.cout, "ks", cw.ks, "dwu", cw.dwu, " bcoef type", type(bc).__name__)
# the raw weight of this conv, from the trained network
W = R.load_trained("trained_fcn16.npz")
cands = [(k, v) for sd in W.values() for k, v in sd.items() if torch.is_tensor(v) and v.dim() == 4 and tuple(v.shape) == (cw.cout, cw.cin, 3, 3)]
print("weight candidates:", [k for k, _ in cands])
pa, pb, pc = ops.coef_ptrs(bc)
res = {}
for tag, fetch in (("direct", 0), ("wino", ops.FETCH_WINOGRAD), ("wino+U", ops.FETCH_WINOGRAD | (ops.FETCH_WINO_U if cw.dwu else 0))):
    o, tab = ops.conv2d_actbwd(g, cw.dwp, cw.cin, 3, u, coef, cap["slope"], pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=u2, fetch=fetch)
    torch.cuda.synchronize(); res[tag] = o.clone()
print("engine's own output vs wino+U alone: equal bits", torch.equal(cap["out"], res["wino+U"]))
bcd = bc.cpu().double()
xin = bcd[:, 0].view(1, -1, 1, 1) * g.cpu().double() + bcd[:, 1].view(1, -1, 1, 1) * u2.cpu().double() + bcd[:, 2].view(1, -1, 1, 1)
c4 = coef.cpu().double()
pre = c4[:, 0].view(1, -1, 1, 1) * u.cpu().double() + c4[:, 1].view(1, -1, 1, 1)
mask = torch.where(pre > 0, 1.0, cap["slope"])
print("prologue'd input: max", float(xin.abs().max()), "mean", float(xin.mean()), "rms", float(xin.pow(2).mean().sqrt()), " g max", float(g.abs().max()), " be*u+de max", float((xin - bcd[:, 0].view(1, -1, 1, 1) * g.cpu().double()).abs().max()))
for k, w in cands:
    ref = F.conv_transpose2d(xin, w.double().cpu(), padding=1) * mask
    line = f"  weight {k}: "
    for tag, o in res.items():
        d = (o.cpu().double() - ref).abs()
        line += f"{tag} max-norm {float(d.max() / ref.abs().max()):.2e} l2 {float(d.norm() / ref.norm()):.2e} | "
    print(line)
    d = (res["wino+U"].cpu().double() - ref).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print("    worst element of wino+U at", idx, "ours", float(res["wino+U"].cpu()[idx]), "ref", float(ref[idx]), "direct", float(res["direct"].cpu()[idx]), " |pre| there", float(pre[idx].abs()))
    big = (d > 1e-4 * ref.abs().max())
    print("    elements off by > 1e-4 of max:", int(big.sum()), "of", d.numel(), "; by image", big.sum((1, 2, 3)).tolist(), "; by row", big.sum((0, 1, 3)).tolist(), "; by col", big.sum((0, 1, 2)).tolist())
    print("    by channel", big.sum((0, 2, 3)).tolist())
