# usage: bash tools/ab_lib.sh   (A/B of maxstyle_amd/lib/alt/libmaxstyle_hip.so against the default build)
ALT=$PWD/maxstyle_amd/lib/alt/libmaxstyle_hip.so
MS_LIB=$ALT python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -2
for f in alt def alt def; do
  if [ $f = alt ]; then export MS_LIB=$ALT; else unset MS_LIB; fi
  python bench.py --steps 300 --warmup 30 --no-parity --no-secondary --no-rccl-selftest --no-outer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$f', round(d['value'],2), round(d['ms_per_step'],4))"
done
