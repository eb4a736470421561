#!/bin/bash
# one gpurun call: everything profiles/r<round>_<tag>_* is made of, except the -m gpu suite (run separately: it alone
# takes 7-14 minutes) -- smoke(), the default bench line, its rocprofv3 kernel summary, the preset table, the PMC passes,
# the wide-payload lines.  Usage on the GPU box: bash scripts/gpu_evidence.sh <tag>
tag=${1:-ev}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --forward-frames 0 > gpurun_out/${tag}_bench_line_20steps.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_prof_bench_line.json 2> gpurun_out/${tag}_prof_bench.err || { tail -20 gpurun_out/${tag}_prof_bench.err; exit 1; }
cp "$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
python - "$tag" <<'PY'
import json,sys
for f in ("bench_line","bench_line_20steps","prof_bench_line"):
    d=json.load(open(f"gpurun_out/{sys.argv[1]}_{f}.json"))
    print(f, round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms; median", round(d["ms_per_step_median"],4), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), "walked", round(d["roofline"]["walked"]["frac"],4), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
# the lines next to the headline: drop-in mode, the reference's evaluation flags (with and without the colour-only forward
# + 4-part-record backward), and where the drop-in step's time goes (rocprofv3 over 20 of its steps)
bash scripts/bench_modes.sh ${tag}_modes > gpurun_out/${tag}_bench_modes.txt 2>&1; cat gpurun_out/${tag}_bench_modes.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_dropin -o p -- python3 bench.py --mode dropin --steps 20 --warmup 3 --no-cpu-baseline --forward-frames 0 > /dev/null 2> gpurun_out/${tag}_prof_dropin.err
cp "$(find gpurun_out/${tag}_prof_dropin -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_dropin_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof_dropin
bash scripts/bench_presets.sh $tag
bash scripts/pmc_collect.sh $tag
for c in 16 64; do python3 bench.py --channels $c --steps 20 --warmup 3 2>/dev/null >> gpurun_out/${tag}_wide_headline.jsonl; python3 bench.py --preset bicycle --channels $c --steps 10 --warmup 2 2>/dev/null >> gpurun_out/${tag}_wide_bicycle.jsonl; done
python3 - "$tag" <<'PY'
import json,sys
for f in ("wide_headline","wide_bicycle"):
    for l in open(f"gpurun_out/{sys.argv[1]}_{f}.jsonl"):
        d=json.loads(l); print(f, d["config"]["channels"], round(d["value"],1), "it/s  K6/K7", d["kernel_ms"]["render_fwd"], d["kernel_ms"]["render_bwd"], "peak", d["hbm_peak_gb"]["allocated"])
PY
