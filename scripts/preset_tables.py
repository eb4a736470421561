"""Markdown tables of a scripts/bench_presets.sh run (bench lines + rocprofv3 kernel stats) for profiles/README.md.
Usage: python scripts/preset_tables.py profiles/r02_v1_presets"""
import csv, json, sys
base = sys.argv[1]
names = ["headline_r6", "headline_r12", "headline_r24", "scan24", "bicycle", "truck"]
FWD = ("preprocess_fwd", "preprocess_color", "color_apply", "rs_", "scan_reduce_kernel<unsigned int", "scan_apply_kernel<unsigned int", "emit_instances",
       "rank_gather", "finalize_bins", "render_fwd")
ADAM = ("adam_",)
groups = [("K6 render_fwd", ("render_fwd",)), ("K7 render_bwd", ("render_bwd",)), ("reduce_rows", ("reduce_rows",)),
          ("K8 preprocess_bwd", ("preprocess_bwd",)), ("K1 + colour (pass or cache apply)", ("preprocess_fwd", "preprocess_color", "color_apply")),
          ("sorts (18 launches)", ("rs_",)), ("emit + gather + finalize", ("emit_instances", "rank_gather", "finalize_bins")),
          ("slot count + scans", ("slot_count", "scan_")), ("Adam (SH + geometry)", ADAM), ("loss + regularizer", ("loss_", "reg_", "objective"))]
rows, ktab = [], {}
for n in names:
    d = json.load(open(f"{base}/{n}_bench_line.json"))
    c = d["config"]
    ks = [(r["Name"].replace("void ", ""), float(r["TotalDurationNs"]), int(r["Calls"])) for r in csv.DictReader(open(f"{base}/{n}_kernel_stats.csv"))]
    n_fwd = max(c_ for k, t, c_ in ks if k.startswith("render_fwd_kernel<0, true"))
    n_bwd = max(c_ for k, t, c_ in ks if k.startswith("render_bwd"))
    n_adam = max(c_ for k, t, c_ in ks if k.startswith("adam_sh"))
    def per_step(prefixes):
        tot = 0.0
        for k, t, c_ in ks:
            if any(k.startswith(p) for p in prefixes):
                ref = n_adam if k.startswith(ADAM) else (n_fwd + 2 if any(k.startswith(p) for p in FWD) else n_bwd)
                tot += t / ref
        return tot / 1e3
    ktab[n] = [per_step(p) for _, p in groups]
    N, D, P = c["gaussians"], c["instances_D"], c["width"] * c["height"]
    rows.append(f"| `{n}` | {N/1e6:g} M @ {c['width']}x{c['height']}, r = {c['radius_px']:g} px | {D/1e6:.2f} M | {c['tile_list_mean']:.0f} | "
                f"{c['entries_walked_per_pixel_mean']:.0f} / {c['entries_walked_per_pixel_max']} | **{d['value']:.1f}** | {d['ms_per_step']:.3f} / "
                f"{d['ms_per_step_median']:.3f} | {d['forward_only_fps']:.0f} | {d['hbm_peak_gb']['allocated']} | {(D*76+P*60)/1e6:.0f} / {(D*148+P*60)/1e6:.0f} |")
print("| preset | scene | D | mean tile list | entries walked per pixel (mean / max) | it/s | ms/step (mean / median) | forward-only FPS | peak HBM GiB | algorithmic MB per launch K6 / K7 |")
print("|---|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
print()
print("| preset (us per step under rocprofv3) | " + " | ".join(g for g, _ in groups) + " | sum |")
print("|---|" + "---|" * (len(groups) + 1))
for n in names:
    print(f"| `{n}` | " + " | ".join(f"{v:.0f}" for v in ktab[n]) + f" | {sum(ktab[n]):.0f} |")
