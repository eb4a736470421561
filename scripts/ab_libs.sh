#!/bin/bash
# Developer aid: same-box A/B of several builds of the library (gaussmart_amd/lib/libgsr_hip_<tag>.so; "cur" = the
# in-tree build), alternating processes.  Usage: bash scripts/ab_libs.sh "cur nopf6 nopf7" [rounds] [bench args...]
tags=$1; rounds=${2:-2}; shift; shift
for i in $(seq 1 $rounds); do
  for t in $tags; do
    if [ $t = cur ]; then unset GSR_LIB_PATH; else export GSR_LIB_PATH=$GRAFT_REPO_ROOT/gaussmart_amd/lib/libgsr_hip_$t.so; fi
    python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --forward-frames 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), 'median', round(d['ms_per_step_median'],4), d['kernel_ms_warmup'])"
  done
done
