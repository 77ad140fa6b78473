# usage: bash tools/ab_switch.sh ENVVAR "pytest -k expr"
python -m pytest tests/test_round3_gpu.py -x -q -k "$2" 2>&1 | tail -6
for f in 1 0 1 0; do env $1=$f python bench.py --steps 300 --warmup 30 --no-parity --no-secondary --no-rccl-selftest --no-outer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1=$f', round(d['value'],2), round(d['ms_per_step'],4))"; done
