#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on this library's access patterns (scripts/microbench/fetch_calib.hip):
# separate --pmc passes, kernel trace only.  Usage on the GPU box: bash scripts/fetch_calib.sh -> gpurun_out/fetch_calib.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fetch_calib
mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- ./scripts/microbench/fetch_calib > $out/bytes.json 2> $out/fetch.log || { tail $out/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- ./scripts/microbench/fetch_calib > /dev/null 2> $out/write.log || { tail $out/write.log; exit 1; }
python3 - <<'PY'
import csv, glob, json
out = "gpurun_out/fetch_calib"
known = json.loads([l for l in open(f"{out}/bytes.json") if l.startswith("{")][-1])
res = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for path in glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != ctr:
                continue
            name = row["Kernel_Name"].split("(")[0]
            res.setdefault(name, {})[ctr + "_kb"] = res.get(name, {}).get(ctr + "_kb", 0.0) + float(row["Counter_Value"])
for k, v in res.items():
    v["known_bytes"] = known.get(k)
json.dump(res, open("gpurun_out/fetch_calib.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
