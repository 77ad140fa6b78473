#!/bin/bash
# Run on the GPU box (gpurun): everything the round's profiles/ entries come from.
#   bash tools/profile_round.sh r02      -> gpurun_out/prof_r02/ ; copy the summaries to keep into profiles/
# rocprofv3 gets the python program directly after `--` (no env / bash -c hop); PMC passes are separate from each other (SQ 8 slots, FETCH_SIZE 3 TCC
# slots, WRITE_SIZE 2) and carry only --kernel-trace.
set -u
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O
# 1. the bench under kernel-trace: per-kernel totals, the launch list of one inner step, the isolated runs bench.py prices; + the per-launch step budget
#    (tools/step_budget.py: launch ledger of one eager step merged with the in-step durations of this trace)
python tools/step_budget.py record c2 $O/ledger_c2.json > $O/ledger.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0 > $O/bench_under_rocprof.json 2> $O/trace.err
python tools/prof_summary.py $O/trace $O/${R}_kernel_stats.txt > /dev/null
python tools/step_budget.py merge $O/ledger_c2.json $O/trace $O/${R}_step_budget_c2 >> $O/ledger.log 2>&1
# 2. config 4 (FCN_64, 16x3x320x320)
python tools/step_budget.py record c4 $O/ledger_c4.json >> $O/ledger.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0 > $O/bench_c4_under_rocprof.json 2> $O/trace_c4.err
python tools/prof_summary.py $O/trace_c4 $O/${R}_c4_kernel_stats.txt > /dev/null
python tools/step_budget.py merge $O/ledger_c4.json $O/trace_c4 $O/${R}_step_budget_c4 >> $O/ledger.log 2>&1
# 3. one trainer iteration (standard pass + inner loop + hard pass + backward + AdamW)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -- python tools/prof_train.py 6 > /dev/null 2> $O/trace_train.err
python tools/prof_by_kernel.py $O/trace_train 0.5 > $O/${R}_train_pass_kernel_stats.txt
# 4. HBM-side traffic of K1 / K2 at layer 4 (16x16x256x256)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_s -- python tools/bench_kernels.py --iters 3 --only L4 > /dev/null 2> $O/pmc_fetch_s.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_s -- python tools/bench_kernels.py --iters 3 --only L4 > /dev/null 2> $O/pmc_write_s.err
python tools/pmc_traffic.py $O/pmc_fetch_s $O/pmc_write_s $O/${R}_traffic_style.json > /dev/null
# 5. SQ / TCC counters of the wide conv kernels at their dominant shapes
bash tools/pmc_conv.sh $R > /dev/null 2>&1
cp gpurun_out/pmc_conv_$R/summary.txt $O/${R}_conv_wide_pmc.txt
python tools/make_traffic.py $O/${R}_traffic_style.json gpurun_out/pmc_conv_$R $O/${R}_traffic.json
# 5b. config 4's dominant shapes: SQ counters + FETCH_SIZE / WRITE_SIZE (tools/pmc_one.sh on tools/trace_conv.py: 64 -> 64 @16x320x320 with the BatchNorm-apply and the
#     two-tensor prologue, two channel blocks per staged tile, weights from the appendix; 256 -> 256 @16x80x80 in the block form; the round-3 kernel (0x500) beside them)
{ for args in "pro1 64 320 0x900" "bwd 64 320 0x900" "pro1 64 320 0x500" "pro1 256 80 0x900" "pro1 256 80 0x500"; do
    echo "== tools/trace_conv.py $args"; bash tools/pmc_one.sh $O/pmc_c4_$(echo $args | tr ' ' '_') $args 2>&1 | grep -v amdgpu; echo; done; } > $O/${R}_conv_wide_pmc_c4.txt
# 5c. the round's parity numbers and the per-layer A/B of the Winograd variants
python tools/parity_report.py r4 > $O/${R}_parity_report.txt 2>&1
python tools/ab_wino_nt.py c2 > $O/${R}_wino_ab_c2.txt 2>&1
python tools/ab_wino_nt.py c4 10 > $O/${R}_wino_ab_c4.txt 2>&1
# 5d. round 4: the sub-pixel and 1x1 kernels against their first generations, the trainer's phases
python tools/ab_subpix.py c4 10 > $O/${R}_subpix_ab_c4.txt 2>&1
python tools/ab_subpix.py c2 10 > $O/${R}_subpix_ab_c2.txt 2>&1
python tools/ab_k1.py c4 10 > $O/${R}_k1_ab_c4.txt 2>&1
python tools/ab_k1.py c2 10 > $O/${R}_k1_ab_c2.txt 2>&1
python tools/train_timeline.py 10 > $O/${R}_train_timeline.txt 2>&1
# 5e. round 4: fabric traffic per kernel of an eager step against the step budget's algorithmic bytes (the screen that found the small-cout kernel's doubled fetch),
#     and that kernel's own counters
bash tools/traffic_by_kernel.sh c2 $O/${R}_step_budget_c2.json > $O/${R}_traffic_by_kernel_c2.txt 2>&1
bash tools/traffic_by_kernel.sh c4 $O/${R}_step_budget_c4.json > $O/${R}_traffic_by_kernel_c4.txt 2>&1
{ python tools/one_small.py c2; python tools/one_small.py c4; bash tools/pmc_cmd.sh $O/pmc_small small_cout tools/one_small.py c2; bash tools/pmc_cache.sh $O/pmc_small2 small_cout tools/one_small.py c2; } 2>&1 | grep -v amdgpu > $O/${R}_small_cout_pmc.txt
# 5f. round 5: the shipped workloads' step budgets, the small-image resampling A/B, the layer-chain probe, configs 4 / 5 against the reference's draws
bash tools/step_budget_shipped.sh > /dev/null 2>&1
for c in acdc192 prostate224; do cp gpurun_out/sb/step_budget_$c.txt $O/${R}_step_budget_$c.txt; cp gpurun_out/sb/step_budget_$c.json $O/${R}_step_budget_$c.json; done
python tools/ab_subpix_small.py 20 2>&1 | grep -v amdgpu > $O/${R}_ab_subpix_small.txt
timeout 300 python tools/chain_probe.py 2>&1 | grep -v amdgpu > $O/${R}_chain_probe.txt; cp gpurun_out/chain_probe.json $O/${R}_chain_probe.json
python tools/draws_report.py 2>&1 | grep -E "^c4|^c5" > $O/${R}_draws_report.txt
python tools/shipped_report.py 2>&1 | grep -E "^acdc|^prostate|per sample" > $O/${R}_shipped_report.txt
# 5g. round 6: the trainer's roofline budget (tools/train_budget.py), the Winograd forms per layer at the shipped shapes, the wide kernel's per-workgroup wall-clock stamps
python tools/train_budget.py record $O/train_ledger.json > $O/train_budget.log 2>&1
rm -rf /tmp/ktt; rocprofv3 --kernel-trace --output-format csv -d /tmp/ktt -- python tools/train_budget.py trace >> $O/train_budget.log 2>&1
python tools/train_budget.py merge $O/train_ledger.json /tmp/ktt $O/${R}_step_budget_train >> $O/train_budget.log 2>&1
rm -rf /tmp/ktt
python tools/ab_wino_nt.py acdc192 10 2>&1 | grep -v amdgpu > $O/${R}_wino_ab_acdc192.txt
python tools/ab_wino_nt.py prostate224 10 2>&1 | grep -v amdgpu > $O/${R}_wino_ab_prostate224.txt
if [ -f maxstyle_amd/lib/trace/libmaxstyle_hip.so ]; then
  for spec in "actbwd 16 256 0x900" "pro1 16 256 0x900" "plain 16 256 0x900" "actbwd 64 320 0x900"; do
    echo "== tools/trace_conv.py $spec"; MS_LIB=$GRAFT_REPO_ROOT/maxstyle_amd/lib/trace/libmaxstyle_hip.so python tools/trace_conv.py $spec 2>&1 | grep -v amdgpu; echo
  done > $O/${R}_conv_wide_stamps.txt
fi
# 6. un-profiled bench lines
python bench.py > $O/${R}_bench_line.json 2> $O/bench.err
python bench.py --config c4 --steps 10 --warmup 2 > $O/${R}_bench_c4.json 2> $O/bench_c4.err
python tools/tune_conv.py 20 > $O/${R}_conv_tuning_table.txt 2>&1
# the raw traces and counter dumps stay on the box (gpurun copies back at most 64 MiB): keep the summaries
rm -rf $O/trace $O/trace_c4 $O/trace_train $O/pmc_* gpurun_out/pmc_conv_$R/*/ gpurun_out/tbk_* gpurun_out/sb/trace* 2>/dev/null
du -sh gpurun_out | tail -1
ls -la $O
