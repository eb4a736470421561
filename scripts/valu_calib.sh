#!/bin/bash
# VALU issue ceiling + SQ_INSTS_VALU calibration.  Builds scripts/microbench/valu_rate.hip, checks
# in the disassembly that every timed loop holds exactly the instructions it claims (no SLP packing), runs it, then runs
# ONE rocprofv3 --pmc SQ_INSTS_VALU pass over it and compares the counter with the known per-launch instruction count.
# Usage on the GPU box: bash scripts/valu_calib.sh  -> gpurun_out/valu_calib.json (+ stdout)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mb=scripts/microbench
out=gpurun_out/valu_calib
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $mb/valu_rate.hip -o $mb/valu_rate || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -S --cuda-device-only $mb/valu_rate.hip -o $out/valu_rate.s || exit 1
echo "== disassembly check (occurrences in the file: the five calibration loops hold 8 each, 8 + 8 for compare + select; the round-4 kinds add 4 + 4 v_fma_f32, 4 v_add_f32_dpp, 8 + 4 v_cndmask_b32, 8 v_cmp_gt_f32)"
for k in v_fma_f32 v_pk_fma_f32 v_cmp_gt_f32 v_cndmask_b32 v_rcp_f32 v_add_f32_dpp; do echo "$k $(grep -c "^\s*$k" $out/valu_rate.s)"; done | tee $out/disasm_counts.txt
echo "== timing"
$mb/valu_rate | tee $out/rates.txt
echo "== SQ_INSTS_VALU"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $out/pmc -o p -- $mb/valu_rate > $out/pmc.log 2>&1 || { tail $out/pmc.log; exit 1; }
python3 - "$out" <<'PY'
import collections, csv, glob, json, re, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()].append(float(r["Counter_Value"]))
expected, rates = {}, {}
names = {"v_fma_f32": 0, "v_pk_fma_f32": 1, "v_cmp+v_cndmask": 2, "v_rcp_f32": 3, "v_add_f32": 4}
for line in open(out + "/rates.txt"):
    m = re.match(r"(\S+(?: dpp)?)\s+([\d.]+) ms\s+([\d.]+) G wave-instr/s\s+clock\s+(\d+) MHz\s+= ([\d.]+) cycles.*in-wave stamps: ([\d.]+); ([\d.]+) at the nominal.*per launch (\d+)", line)
    if m:
        kind = names[m.group(1).split()[0]]
        expected[kind] = float(m.group(8))
        rates[kind] = {"name": m.group(1), "ms": float(m.group(2)), "g_wave_instr_s": float(m.group(3)), "clock_mhz": float(m.group(4)),
                       "cycles_per_instr_per_simd": float(m.group(5)), "cycles_per_instr_per_simd_in_wave_stamps": float(m.group(6)),
                       "cycles_per_instr_per_simd_at_2400mhz": float(m.group(7))}
res = {}
for k, v in sorted(acc.items()):
    kind = int(re.search(r"<(\d)>", k).group(1))
    mean = sum(v) / len(v)
    res[k] = {"sq_insts_valu_per_launch": mean, "loop_valu_expected": expected.get(kind), "ratio": mean / expected[kind] if kind in expected else None, **rates.get(kind, {})}
    print(k, res[k])
json.dump(res, open("gpurun_out/valu_calib.json", "w"), indent=1)
PY
rm -rf $out/pmc
