#!/bin/bash
# The bench lines next to the headline: the drop-in mode (what INTEGRATION.md section 1 delivers
# under a reference-shaped loop) and the reference's evaluation flags (scripts/dtu_eval.py:45), the latter with and without
# the 4-part-record backward (GSR_FLAG_NO_SURFACE_GRAD); the drop-in mode also with the one-line loss swap (gaussmart_amd.loss_utils).  Usage on the GPU box: bash scripts/bench_modes.sh <tag>
tag=${1:-modes}
mkdir -p gpurun_out
common="--no-cpu-baseline --forward-frames 0"
python3 bench.py $common > gpurun_out/${tag}_fused.json 2> gpurun_out/${tag}_fused.err || { tail gpurun_out/${tag}_fused.err; exit 1; }
python3 bench.py $common --mode dropin > gpurun_out/${tag}_dropin.json 2> gpurun_out/${tag}_dropin.err || { tail gpurun_out/${tag}_dropin.err; exit 1; }
python3 bench.py $common --mode dropin --dropin-loss hip > gpurun_out/${tag}_dropin_hiploss.json 2> gpurun_out/${tag}_dropin_hiploss.err || exit 1
python3 bench.py $common --mode dropin --dropin-loss hip --dropin-adam hip > gpurun_out/${tag}_dropin_hiploss_hipadam.json 2> gpurun_out/${tag}_dropin_hiploss_hipadam.err || exit 1
python3 bench.py $common --mode dropin --dropin-loss hip --dropin-adam hip --dropin-render hip > gpurun_out/${tag}_dropin_hiploss_hipadam_hiprender.json 2> gpurun_out/${tag}_dropin_hiploss_hipadam_hiprender.err || exit 1
python3 bench.py $common --eval-flags > gpurun_out/${tag}_evalflags.json 2> gpurun_out/${tag}_evalflags.err || { tail gpurun_out/${tag}_evalflags.err; exit 1; }
GSR_NO_SURFACE_FAST_PATH=0 python3 bench.py $common --eval-flags > gpurun_out/${tag}_evalflags_general_kernel.json 2> gpurun_out/${tag}_evalflags_general_kernel.err || exit 1
python3 bench.py $common --eval-flags --mode dropin > gpurun_out/${tag}_dropin_evalflags.json 2> gpurun_out/${tag}_dropin_evalflags.err || exit 1
python3 - "$tag" <<'PY'
import json, sys
for f in ("fused", "dropin", "dropin_hiploss", "dropin_hiploss_hipadam", "dropin_hiploss_hipadam_hiprender", "evalflags", "evalflags_general_kernel", "dropin_evalflags"):
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{f}.json"))
    print(f"{f:28s} {d['value']:7.1f} it/s  median step {d['ms_per_step_median']:.3f} ms  K6/K7 {d['kernel_ms'].get('render_fwd')} / {d['kernel_ms'].get('render_bwd')} ms  peak {d['hbm_peak_gb']['allocated']} GiB")
PY
