#!/bin/bash
# PMC counters of ONE convolution shape (GPU box): bash tools/pmc_one.sh <outdir> <trace_conv.py args...>   e.g.  gpurun_out/pmc64 pro1 64 320 0x100
# Two SQ passes (8 counters each) + FETCH_SIZE + WRITE_SIZE, each its own rocprofv3 run with --kernel-trace only (no other trace domain).
O=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq1 -- python tools/trace_conv.py "$@" > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2 -- python tools/trace_conv.py "$@" > /dev/null 2> $O/sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python tools/trace_conv.py "$@" > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python tools/trace_conv.py "$@" > /dev/null 2> $O/write.err
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wide" in r["Kernel_Name"] or "conv_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(O + "/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wide" in r["Kernel_Name"] or "conv_mfma" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel-trace duration under --pmc (us):", ["%.1f" % d for d in dur])
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} {sum(v) / len(v):14.4e}   (n={len(v)})")
PY
rm -rf $O/sq1 $O/sq2 $O/fetch $O/write
