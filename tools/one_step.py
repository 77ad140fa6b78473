"""A few EAGER inner-loop steps (no graph) at config 2 / config 4, for per-kernel counters of any launch of the step (tools/pmc_cmd.sh / pmc_cache.sh filter by kernel name):
    python tools/one_step.py [c2|c4] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
net, size = ((4, 1, 4), 256) if cfg == "c2" else ((1, 3, 2), 320)
eng, W, img, lab, styles, z_i, lab_d = bench.build(dev, 16, size, 0, net)
eng.code, eng.labels = z_i, lab_d
eng._prefix_valid = False
for _ in range(steps):
    im = eng.decode(z_i)
    eng.step(im)
torch.cuda.synchronize()
eng.check_errors()
