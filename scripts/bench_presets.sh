#!/bin/bash
# Bench + rocprofv3 kernel stats at the shapes of every BASELINE.json config and a radius sweep of the headline scene.
# Usage on the GPU box: bash scripts/bench_presets.sh <tag>  -> gpurun_out/presets_<tag>/{*.json,*_kernel_stats.csv}
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/presets_$tag
mkdir -p $out
run() {   # name, bench args...
    name=$1; shift
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.log || { tail -5 $out/$name.log; return 1; }
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o p -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --forward-frames 0 "$@" > /dev/null 2> $out/prof_$name.log || { tail -5 $out/prof_$name.log; return 1; }
    cp $(find $out/prof_$name -name '*kernel_stats.csv' | head -1) $out/${name}_kernel_stats.csv
    rm -rf $out/prof_$name
    echo "$name: $(python3 -c "import json;d=json.load(open('$out/$name.json'));print(round(d['value'],1),'it/s D',d['config']['instances_D'],'list',d['config']['tile_list_mean'],'walked',d['config']['entries_walked_per_pixel_mean'],'fwd fps',round(d['forward_only_fps'] or 0,1),'K6/K7 ms',d['kernel_ms_warmup'].get('render_fwd'),d['kernel_ms_warmup'].get('render_bwd'), 'peak GB', d['hbm_peak_gb'])")"
}
run headline_r6
run headline_r12 --radius-px 12
run headline_r24 --radius-px 24
run scan24 --preset scan24
run bicycle --preset bicycle
run truck --preset truck
