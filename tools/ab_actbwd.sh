#!/bin/bash
# A/B of the fused activation-backward epilogue under rocprofv3 (per-launch timeline of one inner step for both settings)
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab_actbwd
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer > $O/fused.json 2> $O/fused.err
python tools/prof_summary.py $O/fused $O/fused.txt > /dev/null
export MS_FUSE_ACTBWD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer > $O/plain.json 2> $O/plain.err
python tools/prof_summary.py $O/plain $O/plain.txt > /dev/null
rm -rf $O/fused $O/plain
