#!/bin/bash
# LDS bank-conflict screen per kernel of an EAGER step (GPU box): bash tools/lds_by_kernel.sh c2 > out.txt
# conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; LDS time = SQ_LDS_IDX_ACTIVE / CUs / 2.4 GHz against the kernel's duration under the counters.
CFG=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lbk_$CFG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $O/sq -- python tools/one_step.py $CFG 3 > /dev/null 2> $O/sq.err
python - "$O" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"ms::", "", n); n = re.sub(r"\(.*$", "", n)
    return n[:60]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(O + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(O + "/sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':60s} {'launches':>8s} {'us':>8s} {'LDS us/CU':>10s} {'conflict':>9s} {'VALU/wave-cyc':>13s} {'VALU:SALU:LDS:MFMA (M instr)':>30s}")
rows = []
for k, c in agg.items():
    n = len(dur[k]); d = sum(dur[k]) / max(n, 1)
    m = lambda name: sum(c.get(name, [0])) / max(len(c.get(name, [0])), 1)
    act, conf = m("SQ_LDS_IDX_ACTIVE"), m("SQ_LDS_BANK_CONFLICT")
    rows.append((act / 256 / 2400.0 / max(d, 1e-9), k, n, d, act / 256 / 2400.0, conf / max(act, 1.0), m("SQ_INSTS_VALU") * 4 / max(m("SQ_WAVE_CYCLES"), 1.0),
                 f"{m('SQ_INSTS_VALU') / 1e6:.2f}:{m('SQ_INSTS_SALU') / 1e6:.2f}:{m('SQ_INSTS_LDS') / 1e6:.2f}:{m('SQ_INSTS_MFMA') / 1e6:.2f}"))
for share, k, n, d, l, cf, vw, mix in sorted(rows, reverse=True):
    if d * n < 15:
        continue
    print(f"{k:60s} {n:8d} {d:8.1f} {l:10.1f} {cf:9.2f} {vw:13.2f} {mix:>30s}")
PY
rm -rf $O/sq
