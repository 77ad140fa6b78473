cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sb; mkdir -p $O
python tools/step_budget.py record c4 $O/ledger_c4.json > $O/ledger.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0 > $O/bench_c4.json 2> $O/trace_c4.err
python tools/step_budget.py merge $O/ledger_c4.json $O/trace_c4 $O/step_budget_c4 >> $O/ledger.log 2>&1
python tools/step_budget.py record c2 $O/ledger_c2.json >> $O/ledger.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0 > $O/bench_c2.json 2> $O/trace.err
python tools/step_budget.py merge $O/ledger_c2.json $O/trace $O/step_budget_c2 >> $O/ledger.log 2>&1
rm -rf $O/trace $O/trace_c4
