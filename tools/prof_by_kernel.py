"""Aggregate a rocprofv3 kernel trace by (trimmed) kernel name: calls, total, average; optionally only the last N-th fraction."""
import csv, glob, os, sys, re

d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4       # drop the warm-up fraction
rows = rows[int(len(rows) * skip):]
agg = {}
for r in rows:
    name = r["Kernel_Name"].replace("void ms::", "").replace("ms::", "")
    name = re.sub(r"\(.*", "", name)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += dur
tot = sum(v[1] for v in agg.values())
print(f"# {len(rows)} launches, {tot/1e3:.2f} ms of kernel time")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[:100]:100s} {c:6d} {t/1e3:9.3f} ms {t/c:8.1f} us {100*t/tot:5.1f}%")
