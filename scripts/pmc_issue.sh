#!/bin/bash
# Where do the wave cycles of the compositing kernels go?  Two rocprofv3 --pmc passes (kernel trace only, SQ counters only)
# over scripts/train_steps_once.py; the SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md):
#   WAVE_CYCLES ~= ACTIVE_INST_ANY + WAIT_INST_ANY (issue stall: dependency / pipe busy) + WAIT_ANY (s_waitcnt / barrier)
# Usage on the GPU box: bash scripts/pmc_issue.sh <tag>  -> gpurun_out/pmc_issue_<tag>.json
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_issue_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/a -o s -- python3 scripts/train_steps_once.py > $out/a.log 2>&1 || { tail $out/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -o s -- python3 scripts/train_steps_once.py > $out/b.log 2>&1 || { tail $out/b.log; exit 1; }
python3 - "$out" "gpurun_out/pmc_issue_${tag}.json" <<'PY'
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
out = {}
for k, cs in sorted(acc.items()):
    o = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    o["avg_us"] = round(sum(dur[k]) / max(len(dur[k]), 1), 1)
    wc = o.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in o:
                o["frac_" + c[3:].lower()] = round(o[c] / wc, 3)
        if o.get("SQ_INSTS_VALU"):
            o["quad_cycles_active_per_valu_inst"] = round(o["SQ_ACTIVE_INST_VALU"] / o["SQ_INSTS_VALU"], 3)
    out[k] = o
json.dump(out, open(dst, "w"), indent=1)
for k in out:
    if "render" in k:
        print(k[:40], out[k])
PY
find $out -name '*.csv' -delete
