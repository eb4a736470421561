#!/bin/bash
# Developer aid: rocprofv3 kernel stats of the radix sort alone (scripts/sort_bench.py) at one shape per run.
# Usage on the GPU box: bash scripts/dev_sort_prof.sh <out dir> "<shape substring>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
mkdir -p $out
for shape in "$@"; do
    tag=$(echo "$shape" | tr ' ' '_')
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o p -- python3 scripts/sort_bench.py "$shape" > $out/$tag.log 2>&1 || { tail -5 $out/$tag.log; exit 1; }
    f=$(find $out/prof_$tag -name '*kernel_stats.csv' | head -1)
    echo "== $shape ($GSR_LIB_PATH)"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("rs_", "fillBuffer", "iota")):
        print(f"{int(r['Calls']):5d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:60]}")
PY
    rm -rf $out/prof_$tag
done
