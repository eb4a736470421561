#!/bin/bash
# HBM traffic and SQ counters per kernel, as MI355X_MICROARCH.md prescribes: SEPARATE rocprofv3 --pmc passes (one
# counter group each, --kernel-trace only) over scripts/train_steps_once.py; scripts/pmc_summarize.py then averages per
# kernel and launch.  Usage (on the GPU box): bash scripts/pmc_collect.sh <tag>   -> gpurun_out/pmc_<tag>_summary.json
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 scripts/train_steps_once.py > $out/fetch.log 2>&1 || { tail $out/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 scripts/train_steps_once.py > $out/write.log 2>&1 || { tail $out/write.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o s -- python3 scripts/train_steps_once.py > $out/sq.log 2>&1 || { tail $out/sq.log; exit 1; }
python3 scripts/pmc_summarize.py $out gpurun_out/pmc_${tag}_summary.json
find $out -name '*kernel_trace.csv' -delete
