"""Kernel trace of tools/prof_call.py -> per call: wall from first to last kernel, sum of kernel time, the largest gaps and what follows them."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# split into calls at gaps > 2 ms (the synchronise between calls)
calls, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 1_500_000:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
for c in calls[-3:]:
    wall = (c[-1][1] - c[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in c) / 1e3
    gaps = sorted(((b[0] - a[1]) / 1e3, a[2][:40], b[2][:40]) for a, b in zip(c, c[1:]))[::-1][:8]
    print(f"call: {len(c)} kernels, wall {wall:.0f} us, kernel time {busy:.0f} us, idle {wall - busy:.0f} us; largest gaps:")
    for g in gaps:
        print(f"    {g[0]:7.1f} us  after {g[1]:40s} before {g[2]}")
