#!/bin/bash
# Kernel trace of the bench's inner step on the GPU box -> gpurun_out/<tag>_kernel_stats.txt (per-launch timeline of one replayed step).
#   bash tools/prof_step.sh <tag> [extra bench.py flags]
set -u
T=${1:-step}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-rccl-selftest --steady-seconds 0 "$@" > $O/bench_under_rocprof.json 2> $O/trace.err
python tools/prof_summary.py $O/trace gpurun_out/${T}_kernel_stats.txt > /dev/null
rm -rf $O/trace
