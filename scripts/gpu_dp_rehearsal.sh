#!/bin/bash
# Rehearsals of bench.py's N > 1 code path on ONE MI355X (the real multi-GPU runs are the driver's):
#   (a) RCCL, world size 1, GSR_BENCH_FORCE_DP=1: every collective, stream, event, the replica check and the exposed-
#       communication probe of the pipelined view-parallel step, with zero link time;
#   (b) gloo, two ranks on the one GPU: the blocking exchange with a real peer (RCCL refuses two ranks per GPU).
# Usage on the GPU box: bash scripts/gpu_dp_rehearsal.sh  -> gpurun_out/dp_rehearsal_{rccl1,gloo2}.json
mkdir -p gpurun_out
GSR_BENCH_FORCE_DP=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 100 --warmup 20 > gpurun_out/dp_rehearsal_rccl1.json 2> gpurun_out/dp_rehearsal_rccl1.log || { tail -20 gpurun_out/dp_rehearsal_rccl1.log; exit 1; }
GSR_BENCH_BACKEND=gloo GSR_BENCH_SINGLE_DEVICE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
    bench.py --gpus 2 --steps 30 --warmup 5 --gaussians 300000 > gpurun_out/dp_rehearsal_gloo2.json 2> gpurun_out/dp_rehearsal_gloo2.log || { tail -20 gpurun_out/dp_rehearsal_gloo2.log; exit 1; }
python3 - <<'PY'
import json
for f in ("rccl1", "gloo2"):
    d = json.load(open(f"gpurun_out/dp_rehearsal_{f}.json"))
    print(f, round(d["value"], 1), "it/s", d["n_gpus"], "ranks;", d.get("view_parallel"))
PY
