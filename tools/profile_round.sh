#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace statistics of the bench + HBM-traffic PMC passes of the two priced kernels.
# Usage: bash tools/profile_round.sh r01
set -u
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.err
python tools/prof_summary.py $O/trace $O/${R}_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python tools/bench_conv.py c16_256 3 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python tools/bench_conv.py c16_256 3 > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/${R}_traffic_conv.json > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_s -- python tools/bench_kernels.py --iters 3 --only L4 > /dev/null 2> $O/pmc_fetch_s.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_s -- python tools/bench_kernels.py --iters 3 --only L4 > /dev/null 2> $O/pmc_write_s.err
python tools/pmc_traffic.py $O/pmc_fetch_s $O/pmc_write_s $O/${R}_traffic_style.json > /dev/null
ls -la $O
