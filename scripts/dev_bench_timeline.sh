#!/bin/bash
# Developer aid: kernel timeline of ONE pipelined training step of bench.py (both streams), from a rocprofv3 kernel trace.
# Usage on the GPU box: bash scripts/dev_bench_timeline.sh [bench args...]  -> stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/bench_timeline
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --forward-frames 0 "$@" > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""),
                     r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
k7 = [i for i, r in enumerate(rows) if r[2].startswith("render_bwd_kernel")]
a, b = k7[8], k7[9]          # inside the timed region
t0 = rows[a][0]
qs = sorted({r[3] for r in rows[a:b + 1]})
print("queues:", qs)
for s, e, n, q, st in rows[a:b + 1]:
    col = qs.index(q)
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {'    ' * col}[q{col}] {n[:60]}")
print("step span (K7 start to next K7 start)", (rows[b][0] - rows[a][0]) / 1e3, "us")
PY
find $out -name '*.csv' -delete
