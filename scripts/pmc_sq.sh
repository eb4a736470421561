#!/bin/bash
# SQ counters only (one rocprofv3 --pmc pass, kernel trace only) over scripts/train_steps_once.py.
# Usage on the GPU box: bash scripts/pmc_sq.sh <tag>  -> gpurun_out/pmc_sq_<tag>.json
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_sq_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o s -- python3 scripts/train_steps_once.py > $out/sq.log 2>&1 || { tail $out/sq.log; exit 1; }
python3 - "$out" "gpurun_out/pmc_sq_${tag}.json" <<'PY'
import collections, csv, glob, json, re, sys
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} | {"avg_us": round(sum(dur[k]) / max(len(dur[k]), 1), 1)} for k, cs in sorted(acc.items())}
json.dump(out, open(dst, "w"), indent=1)
for k in out:
    if "render" in k or "reduce_rows" in k:
        print(k[:40], out[k])
PY
find $out -name '*.csv' -delete
