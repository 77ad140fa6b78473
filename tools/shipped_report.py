import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import r5_cases as R5
from maxstyle_amd.options import engine_defaults
dev = torch.device("cuda:0")
for which in ("acdc", "prostate"):
    for w in (True, False):
        with engine_defaults(winograd=w):
            r = R5.shipped_case(dev, which)
        print(which, "winograd" if w else "direct", json.dumps({k: (v if not isinstance(v, float) else float("%.3e" % v)) for k, v in r.items() if "per_sample" not in k}))
        print("   per sample max:", " ".join("%.1e" % v for v in r["image_max_per_sample"]))
        print("   per sample rms:", " ".join("%.1e" % v for v in r["image_rms_per_sample"]))
        print("   reference draws, per sample max:", {k: [" ".join("%.1e" % x for x in row) for row in v] for k, v in r["draws"].items() if "per_sample" in k})
