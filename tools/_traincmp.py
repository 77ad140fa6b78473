import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
rows = []
for k in a:
    if k in b and a[k].shape == b[k].shape:
        den = float(b[k].abs().max()) or 1.0
        rows.append((float((a[k] - b[k]).abs().max()) / den, k))
rows.sort(reverse=True)
for r in rows[:25]: print("%.3e  %s" % r)
