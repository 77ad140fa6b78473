#!/bin/bash
# PMC counters of the kernels whose name contains <substr>, for any python command (GPU box):
#   bash tools/pmc_cmd.sh <outdir> <substr> <python script and args...>      e.g.  gpurun_out/pmc_sub conv_subpix tools/ab_subpix.py c4 5 0
# Two SQ passes (8 counters each) + FETCH_SIZE + WRITE_SIZE, each its own rocprofv3 run with --kernel-trace only (no other trace domain); python is started directly.
O=$1; K=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq1 -- python "$@" > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2 -- python "$@" > /dev/null 2> $O/sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python "$@" > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python "$@" > /dev/null 2> $O/write.err
python - "$O" "$K" <<'PY'
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(O + "/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
dur.sort()
print("kernel:", K, " launches:", len(dur), " median duration under --pmc (us): %.1f" % (dur[len(dur) // 2] if dur else 0.0))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} {sum(v) / len(v):14.4e}   (n={len(v)})")
PY
rm -rf $O/sq1 $O/sq2 $O/fetch $O/write
