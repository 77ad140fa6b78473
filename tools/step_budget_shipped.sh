# per-launch step budget of the reference's SHIPPED workloads (20x1x192x192 ACDC, 20x1x224x224 Prostate): ledger + kernel trace -> gpurun_out/sb/step_budget_<cfg>.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sb; mkdir -p $O
for cfg in "acdc192 192 4" "prostate224 224 2"; do
  set -- $cfg
  python tools/step_budget.py record $1 $O/ledger_$1.json >> $O/ledger.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$1 -- python bench.py --batch 20 --size $2 --classes $3 --steps 10 --warmup 2 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0 > $O/bench_$1.json 2> $O/trace_$1.err
  python tools/step_budget.py merge $O/ledger_$1.json $O/trace_$1 $O/step_budget_$1 >> $O/ledger.log 2>&1
  rm -rf $O/trace_$1
done
