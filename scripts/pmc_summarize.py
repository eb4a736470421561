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
