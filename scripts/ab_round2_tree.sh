#!/bin/bash
# Developer aid: same-box comparison of THIS tree with a checkout of an earlier round's tree (Python + library of that
# round, built in place), alternating processes.  The old tree is not part of the repository:
#   mkdir .ab_r02 && git archive <round-2 commit> | tar -x -C .ab_r02 && make -C .ab_r02/gaussmart_amd/csrc
# Usage on the GPU box: bash scripts/ab_round2_tree.sh [rounds] [bench args...]
rounds=${1:-3}; shift
for i in $(seq 1 $rounds); do
  for t in . .ab_r02; do
    (cd $GRAFT_REPO_ROOT/$t && python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --forward-frames 0 "$@" 2>/dev/null) | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), 'it/s  median', round(d['ms_per_step_median'],4), 'ms', d.get('kernel_ms_warmup'))"
  done
done
