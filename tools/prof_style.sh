#!/bin/bash
# Kernel-trace durations (not host-paced event loops) of the MaxStyle kernels per geometry / cache-policy setting.
# Usage (GPU box): bash tools/prof_style.sh > gpurun_out/style_prof.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "512 0" "512 3" "512 1" "512 2" "0 3"; do
  set -- $cfg
  export MS_STYLE_FUSED_THREADS=$1 MS_STYLE_FUSED_NT=$2
  O=gpurun_out/prof_style_$1_$2
  rm -rf $O
  rocprofv3 --kernel-trace --output-format csv -d $O -- python tools/bench_kernels.py --iters 20 > /dev/null 2> $O.err
  echo "== MS_STYLE_FUSED_THREADS=$1 MS_STYLE_FUSED_NT=$2"
  python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "style" not in n and "moments" not in n:
        continue
    key = (n.split("(")[0].replace("void ms::", ""), r["Grid_Size_X"], r["Workgroup_Size_X"])
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print("%-44s grid %8s wg %5s  n=%3d  median %7.1f us  min %7.1f" % (k[0][:44], k[1], k[2], len(v), v[len(v) // 2], v[0]))
PY
done
