#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace statistics of the bench and of the training pass + HBM-traffic PMC passes of the priced kernels.
# Usage: bash tools/profile_round.sh r01     (summaries land in gpurun_out/prof_<round>/; copy the ones to keep into profiles/)
set -u
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer > $O/bench_under_rocprof.json 2> $O/trace.err
python tools/prof_summary.py $O/trace $O/${R}_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -- python tools/prof_train.py 6 > /dev/null 2> $O/trace_train.err
python tools/prof_by_kernel.py $O/trace_train 0.5 > $O/${R}_train_pass_kernel_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python tools/bench_conv.py c16_256 3 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python tools/bench_conv.py c16_256 3 > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/${R}_traffic_conv.json > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_s -- python tools/bench_kernels.py --iters 3 --only L4 > /dev/null 2> $O/pmc_fetch_s.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_s -- python tools/bench_kernels.py --iters 3 --only L4 > /dev/null 2> $O/pmc_write_s.err
python tools/pmc_traffic.py $O/pmc_fetch_s $O/pmc_write_s $O/${R}_traffic_style.json > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_w -- python tools/bench_wgrad.py u4.c0 > /dev/null 2> $O/pmc_fetch_w.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_w -- python tools/bench_wgrad.py u4.c0 > /dev/null 2> $O/pmc_write_w.err
python tools/pmc_traffic.py $O/pmc_fetch_w $O/pmc_write_w $O/${R}_traffic_wgrad.json > /dev/null
python tools/bench_wgrad.py > $O/${R}_wgrad_bench.txt 2>&1
python tools/bench_conv.py all 30 > $O/${R}_conv_bench.txt 2>&1
python tools/bench_train.py > $O/${R}_train_iteration.json 2> /dev/null
ls -la $O
