#!/bin/bash
# Developer aid: kernel trace of a few training steps; prints the sort / scan kernels of ONE iteration in launch order.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_sort
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 scripts/train_steps_once.py > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
rows.sort()
# last full iteration: from the last preprocess_fwd_kernel to the end
starts = [i for i, r in enumerate(rows) if r[2].startswith("preprocess_fwd_kernel")]
a = starts[-2]; b = starts[-1]
prev_end = rows[a][0]
tot = 0
for s, e, n in rows[a:b]:
    print(f"{(s - rows[a][0]) / 1e3:9.1f} us  +gap {max(0, s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {n[:70]}")
    prev_end = max(prev_end, e)
print("iteration span", (rows[b][0] - rows[a][0]) / 1e3, "us")
PY
find $out -name '*.csv' -delete
