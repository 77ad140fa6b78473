"""Summarise tools/pmc_conv.sh: per replayed variant, counters of the LAST 10 dispatches of the wide conv kernel (the replays), averaged per launch."""
import csv, glob, os, sys, collections


def last_dispatches(d, n=10):
    fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        return {}
    rows = [r for r in csv.DictReader(open(fs[0])) if "conv_wide_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-n:]
    acc = collections.defaultdict(list)
    name = None
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"].split("(")[0].replace("void ms::", "")
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    out["_kernel"] = name
    return out


def dur(d, n=10):
    fs = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    if not fs:
        return None
    rows = [r for r in csv.DictReader(open(fs[0])) if "conv_wide_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-n:]]
    return sum(ds) / len(ds)


def main(root):
    print("# rocprofv3 --pmc on tools/replay_conv.py (10 back-to-back replays of one launch of the C2 step on its live buffers); per-launch averages")
    print("# SQ_* cycle counters are summed over all waves (quad-cycles for WAVE_CYCLES / WAIT_* / ACTIVE_INST_*; cycles for VALU_MFMA_BUSY_CYCLES: MI355X_MICROARCH.md)")
    print("# FETCH_SIZE doubled (gfx950 counts 128-B requests of 16-B/lane streaming reads as 64 B), WRITE_SIZE as reported; both KiB -> MB")
    for w in ("dgrad_actbwd", "dgrad_plain", "dgrad_acc", "dgrad_nt2", "fwd_pro1", "fwd_pro0"):
        a = last_dispatches(os.path.join(root, "sq_" + w))
        b = last_dispatches(os.path.join(root, "sq2_" + w))
        f = last_dispatches(os.path.join(root, "fetch_" + w))
        wr = last_dispatches(os.path.join(root, "write_" + w))
        if not a:
            print(w, "missing"); continue
        t = dur(os.path.join(root, "fetch_" + w))
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        print(f"\n== {w}: {a.get('_kernel')}   kernel-trace duration under --pmc {t:.1f} us")
        if wc:
            print(f"   wave-cycles {wc:.3e} | WAIT_ANY {a.get('SQ_WAIT_ANY', 0) / wc:6.1%} | WAIT_INST_ANY {a.get('SQ_WAIT_INST_ANY', 0) / wc:6.1%} | ACTIVE_INST_ANY {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1%}")
        busy = a.get("SQ_BUSY_CYCLES", 0.0)
        mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        print(f"   SQ_BUSY_CYCLES {busy:.3e} | SQ_VALU_MFMA_BUSY_CYCLES {mf:.3e} | LDS_BANK_CONFLICT {a.get('SQ_LDS_BANK_CONFLICT', 0):.3e} of LDS_IDX_ACTIVE {a.get('SQ_LDS_IDX_ACTIVE', 0):.3e}")
        if b:
            print("   instructions per launch: " + " ".join(f"{k[8:]} {v:.3e}" for k, v in sorted(b.items()) if k.startswith("SQ_INSTS")) +
                  f" | WAIT_INST_LDS {b.get('SQ_WAIT_INST_LDS', 0):.3e} ACTIVE_INST_VALU {b.get('SQ_ACTIVE_INST_VALU', 0):.3e}")
            nm = b.get("SQ_INSTS_MFMA", 0.0)
            if nm and t:
                # 16x16x4 f32 MFMA: 32 cycles of one SIMD's matrix pipe each; 1024 SIMDs
                print(f"   MFMA pipe time = {nm:.3e} x 32 cyc / 1024 SIMDs = {nm * 32 / 1024:.3e} cycles per SIMD = {nm * 32 / 1024 / 2.4e3:.1f} us at 2.4 GHz -> {nm * 32 / 1024 / 2.4e3 / t:5.1%} of the launch")
        if f and wr:
            rd = f.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6
            wb = wr.get("WRITE_SIZE", 0) * 1024 / 1e6
            print(f"   HBM-side traffic: read {rd:.1f} MB + write {wb:.1f} MB = {rd + wb:.1f} MB per launch")


if __name__ == "__main__":
    main(sys.argv[1])
