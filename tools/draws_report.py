"""Config 4 / config 5 calls on the GPU against the reference's fp64 runs, next to the reference's own fp32 evaluations (tests/golden/loop_ref_draws.npz).  (GPU box.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import r5_cases as R5, r4_cases as R4
from maxstyle_amd.options import engine_defaults
dev = torch.device("cuda:0")
f = lambda v: [float("%.2e" % x) for x in v] if isinstance(v, (list, tuple)) and v and isinstance(v[0], float) else (float("%.2e" % v) if isinstance(v, float) else v)
for w in (True, False):
    with engine_defaults(winograd=w):
        for tag in ("acdc", "prostate"):
            r = R5.c5_call_vs_f64(dev, tag)
            print("c5", tag, "winograd" if w else "direct", json.dumps({k: f(v) for k, v in r.items() if k != "variants"}))
        r = R4.c4_full_case(dev)
        print("c4", "winograd" if w else "direct", json.dumps({k: f(v) for k, v in r.items() if not isinstance(v, dict)}))
print("c4 draws", json.dumps({k: f(v) for k, v in R5.c4_draws().items() if "per_sample" not in k and k != "losses"}))
