#!/bin/bash
# L2 hit / miss and request counters of the kernels whose name contains <substr> (GPU box): bash tools/pmc_cache.sh <outdir> <substr> <python script and args...>
O=$1; K=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p $O
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/c1 -- python "$@" > /dev/null 2> $O/c1.err
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/c2 -- python "$@" > /dev/null 2> $O/c2.err
python - "$O" "$K" <<'PY'
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} {sum(v) / len(v):14.4e}   (n={len(v)})")
PY

rm -rf $O/c1 $O/c2
