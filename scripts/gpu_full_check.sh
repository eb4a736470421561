#!/bin/bash
# one gpurun call: the whole -m gpu suite, smoke(), the default bench line, and the rocprofv3 kernel summary of the
# same command (outputs under gpurun_out/<tag>_*)
tag=${1:-full}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gpu_tests.log 2>&1 || { tail -40 gpurun_out/${tag}_gpu_tests.log; exit 1; }
tail -2 gpurun_out/${tag}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python3 bench.py > gpurun_out/${tag}_prof_bench_line.json 2> gpurun_out/${tag}_prof_bench.err || { tail -20 gpurun_out/${tag}_prof_bench.err; exit 1; }
cp "$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
find gpurun_out/${tag}_prof -name '*kernel_trace.csv' -delete
python - "$tag" <<'PY'
import json,sys
for f in ("bench_line","prof_bench_line"):
    d=json.load(open(f"gpurun_out/{sys.argv[1]}_{f}.json"))
    print(f, round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms; roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
