#!/bin/bash
# Same-box A/B of one environment switch of bench.py (alternating processes).  Usage on the GPU box:
#   bash scripts/ab_env.sh VAR VALUE_A VALUE_B [rounds] [bench args...]
var=$1; a=$2; b=$3; rounds=${4:-3}; shift 4
for i in $(seq 1 $rounds); do
  for val in "$a" "$b"; do
    env $var=$val python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --forward-frames 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$var=$val', round(d['value'],1), 'median', round(d['ms_per_step_median'],4), d['kernel_ms_warmup'])"
  done
done
