#!/bin/bash
# developer aid: time library variants (gaussmart_amd/lib/libgsr_v_*.so) with the same bench
cd "$(dirname "$0")/.."
cp gaussmart_amd/lib/libgsr_hip.so /tmp/libgsr_orig.so
for v in gaussmart_amd/lib/libgsr_v_*.so; do
  cp "$v" gaussmart_amd/lib/libgsr_hip.so
  echo "== $v"
  python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['kernel_ms'])"
done
cp /tmp/libgsr_orig.so gaussmart_amd/lib/libgsr_hip.so
echo "== base"
python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['kernel_ms'])"
