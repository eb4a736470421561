"""Per-kernel, per-launch averages of the rocprofv3 --pmc passes collected by scripts/pmc_collect.sh.
Units as reported by rocprofv3 (FETCH_SIZE / WRITE_SIZE in KB); bench.py applies the gfx950 FETCH_SIZE x 2 correction."""
import collections, csv, glob, json, re, sys
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if name.startswith(("Cijk", "at::", "__amd", "rocsolver", "void at")):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} | {"launches_seen": max(len(v) for v in cs.values())}
       for k, cs in sorted(acc.items())}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        print(f"{k[:44]:44s} fetch {v['FETCH_SIZE']/1024:8.1f} MB (x2 = {2*v['FETCH_SIZE']/1024:8.1f})  write {v['WRITE_SIZE']/1024:8.1f} MB  VALU {v.get('SQ_INSTS_VALU',0):.3g}")

# bench.py's view of the same data: one entry per kernel family (largest instantiation), short names
traffic = {"_comment": "per-launch averages of rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ passes (separate passes, "
                       "scripts/pmc_collect.sh) over scripts/train_steps_once.py at 1M Gaussians / 1920x1080; units as reported "
                       "(KB); bench.py applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x 2) and reports "
                       "2*fetch + write as roofline.traffic; valu_insts / salu_insts = SQ_INSTS_VALU / SQ_INSTS_SALU "
                       "wave-instructions per launch",
           "source": dst, "kernels": {}}
for k, v in out.items():
    if not k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    short = re.sub(r"_kernel.*", "", k)
    cur = traffic["kernels"].get(short)
    if cur and cur["fetch_kb"] + cur["write_kb"] >= v["FETCH_SIZE"] + v["WRITE_SIZE"]:
        continue
    traffic["kernels"][short] = {"fetch_kb": v["FETCH_SIZE"], "write_kb": v["WRITE_SIZE"],
                                 "valu_insts": v.get("SQ_INSTS_VALU"), "salu_insts": v.get("SQ_INSTS_SALU")}
json.dump(traffic, open(dst.replace("_summary.json", "_traffic.json"), "w"), indent=1)
