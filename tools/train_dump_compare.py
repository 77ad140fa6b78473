# (round 3 diagnostic; see profiles/r03_experiments.txt 17-18 and DESIGN.md section 4)
import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in sorted(a):
    if not k.startswith("buf:"): continue
    if k in b and a[k].shape == b[k].shape:
        den = float(b[k].abs().max()) or 1.0
        d = (a[k] - b[k]).abs()
        flips = int(((a[k] > 0) != (b[k] > 0)).sum())
        print("%-28s max %.2e  rms %.2e  sign flips %d / %d" % (k[4:], float(d.max()) / den, float(d.pow(2).mean().sqrt()) / den, flips, a[k].numel()))
