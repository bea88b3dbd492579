#!/bin/bash
# round 3, GPU call 2: per-shape tables of both UNet stream modes (same box)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for m in f16 f32; do
  UAV_BENCH_DETAIL=1 timeout 600 python bench.py --steps 1 --no-cpu-baseline --unet-stream $m > gpurun_out/r3_detail_$m.json 2> gpurun_out/r3_detail_$m.txt
done
UAV_BRANCH_F32=0 UAV_BENCH_DETAIL=1 timeout 600 python bench.py --steps 1 --no-cpu-baseline --unet-stream f32 > gpurun_out/r3_detail_f32_branch16.json 2> gpurun_out/r3_detail_f32_branch16.txt
python - <<'PY'
import json
for m in ("f16", "f32", "f32_branch16"):
    d = json.load(open(f"gpurun_out/r3_detail_{m}.json"))
    print(m, d["value"], d["ms_per_step"], d["roofline"]["achieved"])
PY
