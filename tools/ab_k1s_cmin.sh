F="--no-cpu-baseline --no-outer --no-parity --no-secondary --no-instep --no-rccl-selftest --steady-seconds 0"
for i in 1 2 3; do
  for v in 64 16; do
    r=$(MS_K1S_CMIN=$v python bench.py --steps 200 --warmup 10 $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "cmin $v  $r"
  done
done
