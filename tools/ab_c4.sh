#!/bin/bash
# C4 A/B of the Winograd dispatch switches (GPU box)
for cb in 1000 8 4 2; do
  echo -n "maxcb $cb: "; MS_CONV_WINO_MAXCB=$cb python bench.py --config c4 --no-cpu-baseline --no-outer --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['loss_check'])"
done
echo -n "no w32: "; MS_CONV_WINO32=0 python bench.py --config c4 --no-cpu-baseline --no-outer --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
echo -n "no wino: "; MS_LOOP_WINOGRAD=0 python bench.py --config c4 --no-cpu-baseline --no-outer --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
