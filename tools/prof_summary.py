"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and the per-launch timeline of one inner step."""
import csv, glob, sys, os


def main(d, out=None):
    ks = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    lines = []
    rows = list(csv.DictReader(open(ks)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines.append(f"# kernel stats ({os.path.basename(ks)}), total {tot/1e6:.2f} ms")
    lines.append(f"{'kernel':80s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>6s}")
    for r in rows[:40]:
        lines.append(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.1f}")
    rows = list(csv.DictReader(open(kt)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    inc = [i for i, r in enumerate(rows) if "incr_kernel" in r["Kernel_Name"] or "step_tail_kernel" in r["Kernel_Name"]]      # the last launch of a step
    if len(inc) >= 3:
        a, b = inc[-3] + 1, inc[-2] + 1
        lines.append("")
        lines.append(f"# one inner step (graph replay): {b-a} launches, wall {(int(rows[b-1]['End_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3:.1f} us")
        agg = {}
        for r in rows[a:b]:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            name = r["Kernel_Name"].replace("void ms::", "").replace("ms::", "")
            name = name[:name.index("(")] if "(" in name else name
            lines.append(f"{name[:60]:60s} grid=({int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6d},{r['Grid_Size_Y']:>4s},{r['Grid_Size_Z']:>3s}) vgpr={r['VGPR_Count']:>4s} lds={r['LDS_Block_Size']:>6s} {dur:8.1f} us")
            agg[name] = agg.get(name, 0) + dur
        lines.append("")
        lines.append("# per-step totals by kernel")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
            lines.append(f"{k[:70]:70s} {v:9.1f} us")
    # the launches bench.py prices for its `roofline` objects: back-to-back runs (>= 20) of ONE kernel on ONE shape (kernel_rooflines())
    lines.append("")
    lines.append("# isolated back-to-back runs (bench.py kernel_rooflines: 3 warm-up + 20 timed launches of one shape) - compare with roofline.us_per_launch")
    i = 0
    while i < len(rows):
        j = i
        while j + 1 < len(rows) and rows[j + 1]["Kernel_Name"] == rows[i]["Kernel_Name"]:
            j += 1
        if j - i + 1 >= 20 and "ms::" in rows[i]["Kernel_Name"]:
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:j + 1]][-20:]
            name = rows[i]["Kernel_Name"].replace("void ms::", "").replace("ms::", "")
            name = name[:name.index("(")] if "(" in name else name
            lines.append(f"{name[:60]:60s} n={j - i + 1:3d}  avg of the last 20: {sum(durs) / len(durs):7.1f} us  (min {min(durs):.1f}, max {max(durs):.1f})")
        i = j + 1
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
