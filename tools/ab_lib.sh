#!/bin/bash
# A/B of library builds on the inner step (GPU box): bash tools/ab_lib.sh <label>=<path to .so | default> ...   -> steps/s at config 2 and config 4 per build
F="--no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0"
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset MS_LIB; else export MS_LIB=$lib; fi
  c2=$(python bench.py --steps 20 --warmup 3 $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  c4=$(python bench.py --config c4 --steps 6 --warmup 1 $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "$label  c2 $c2  c4 $c4"
done
