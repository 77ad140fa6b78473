#!/bin/bash
# Kernel trace of the bench's inner step on the GPU box -> gpurun_out/<tag>_kernel_stats.txt (per-launch timeline of one replayed step).
#   bash tools/prof_step.sh <tag> [extra bench.py flags]
set -u
T=${1:-step}; shift
# one rank, profiled directly: with --gpus N / --force-dist bench.py's launcher parent would start rank processes from a process that already carries the profiler's
# preloaded library (under --pmc the GPU is initialised there): the forbidden exec hop
for a in "$@"; do case "$a" in --gpus|--gpus=*|--force-dist) echo "prof_step.sh: profile one rank directly (no --gpus / --force-dist)" >&2; exit 2;; esac; done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0 "$@" > $O/bench_under_rocprof.json 2> $O/trace.err
python tools/prof_summary.py $O/trace gpurun_out/${T}_kernel_stats.txt > /dev/null
# the per-launch step budget (tools/step_budget.py): a launch ledger recorded beforehand as gpurun_out/ledger_<tag>.json is merged with this trace's in-step durations
if [ -f gpurun_out/ledger_$T.json ]; then python tools/step_budget.py merge gpurun_out/ledger_$T.json $O/trace gpurun_out/${T}_step_budget; fi
rm -rf $O/trace
