#!/bin/bash
# Same-box A/B of two builds of libgsr_hip.so: gaussmart_amd/lib/libgsr_hip_base.so (baseline) vs the current one,
# alternating processes, headline bench; prints it/s and the big kernels' event-timed averages of the warm-up.
# Usage on the GPU box: bash scripts/ab_builds.sh [rounds] [bench args...]
rounds=${1:-3}; shift
for i in $(seq 1 $rounds); do
  for which in base new; do
    if [ $which = base ]; then export GSR_LIB_PATH=$GRAFT_REPO_ROOT/gaussmart_amd/lib/libgsr_hip_base.so; else unset GSR_LIB_PATH; fi
    python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --forward-frames 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which', round(d['value'],1), 'median', round(d['ms_per_step_median'],4), d['kernel_ms_warmup'])"
  done
done
