#!/bin/bash
# A/B of library builds on the config-2 step only (GPU box): bash tools/ab_c2.sh <rounds> <label>=<path to .so | default> ...   -> steps/s per build, alternating
F="--no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0"
R=$1; shift
for i in $(seq $R); do
  for spec in "$@"; do
    label=${spec%%=*}; lib=${spec#*=}
    if [ "$lib" = "default" ]; then unset MS_LIB; else export MS_LIB=$lib; fi
    c2=$(python bench.py --steps 100 --warmup 5 $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$label  c2 $c2"
  done
done
