#!/bin/bash
# size-1 RCCL rehearsal of the view-parallel bench step on one GPU (every collective, stream and event of the N > 1 path)
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 GSR_BENCH_FORCE_DP=1
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dp rehearsal', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms')"
