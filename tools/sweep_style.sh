#!/bin/bash
# K1 (single-read MaxStyle forward) geometry / cache-policy sweep on the GPU box: one process per setting (the switches are read once).
# Usage: bash tools/sweep_style.sh > gpurun_out/style_sweep.txt
cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "0 1" "0 2" "0 3" "512 0" "256 0" "512 3"; do
  set -- $cfg
  echo "== MS_STYLE_FUSED_THREADS=$1 MS_STYLE_FUSED_NT=$2"
  MS_STYLE_FUSED_THREADS=$1 MS_STYLE_FUSED_NT=$2 python tools/bench_kernels.py --iters 50 2>gpurun_out/sweep_err.txt | python -c "
import sys, json
d = json.load(sys.stdin)
print('copy GB/s %.0f' % d.pop('copy_256MB_GBps'))
for k, v in d.items():
    print('%-6s fwd %6.1f us %6.0f GB/s | bwd+dx %6.1f us %6.0f GB/s' % (k, v['fwd_us'], v['fwd_GBps'], v['bwd_dx_us'], v['bwd_dx_GBps']))
"
done
echo "== three-launch path (MS_STYLE_FUSED=0)"
MS_STYLE_FUSED=0 python tools/bench_kernels.py --iters 50 2>gpurun_out/sweep_err.txt | python -c "
import sys, json
d = json.load(sys.stdin)
d.pop('copy_256MB_GBps')
for k, v in d.items():
    print('%-6s fwd %6.1f us %6.0f GB/s' % (k, v['fwd_us'], v['fwd_GBps']))
"
