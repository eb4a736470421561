#!/bin/bash
# one gpurun call: parity tests of the factored SH gradient, then A/B bench lines (single GPU and the
# size-1 RCCL rehearsal of the view-parallel step)
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_factored_sh.py tests/test_gpu_view_parallel.py tests/test_gpu_train.py tests/test_gpu_adam.py -x -q > gpurun_out/fs_test.log 2>&1 || { tail -40 gpurun_out/fs_test.log; exit 1; }
tail -3 gpurun_out/fs_test.log
python bench.py --no-cpu-baseline > gpurun_out/fs_bench_factored.json 2> gpurun_out/fs_bench_factored.err
GSR_BENCH_UNFACTORED=1 python bench.py --no-cpu-baseline > gpurun_out/fs_bench_unfactored.json 2> gpurun_out/fs_bench_unfactored.err
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 GSR_BENCH_FORCE_DP=1
python bench.py --no-cpu-baseline > gpurun_out/fs_dp_factored.json 2> gpurun_out/fs_dp_factored.err
GSR_BENCH_UNFACTORED=1 python bench.py --no-cpu-baseline > gpurun_out/fs_dp_unfactored.json 2> gpurun_out/fs_dp_unfactored.err
for f in fs_bench_factored fs_bench_unfactored fs_dp_factored fs_dp_unfactored; do
  python - "$f" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms", d["kernel_ms_warmup"])
PY
done
