#!/bin/bash
# Developer aid: rocprofv3 per-kernel averages of bench.py for the build GSR_LIB_PATH selects, restricted to the kernels
# whose name contains one of the given substrings.  Usage on the GPU box: bash scripts/dev_kstats.sh <tag> sub1 sub2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --forward-frames 0 > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
f=$(find $out -name '*kernel_stats.csv' | head -1)
echo "== $tag ($GSR_LIB_PATH)"
python3 - "$f" "$@" <<'PY'
import csv, sys
subs = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    if not subs or any(k in r["Name"] for k in subs):
        print(f"{int(r['Calls']):5d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:60]}")
PY
find $out -name '*.csv' -delete
