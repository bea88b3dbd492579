#!/bin/bash
# round 3, GPU call 9: LayerNorm change (gamma/beta hoisted, next row prefetched) tests + full per-shape table (90 rows) of the default
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_clip_text_gpu.py -q -m gpu -x 2>&1 | tail -3
UAV_BENCH_DETAIL=1 timeout 600 python bench.py --steps 1 --no-cpu-baseline > gpurun_out/r3_detail_default_d.json 2> gpurun_out/r3_detail_default_d.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_detail_default_d.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k: v["ms"] for k, v in d["kernel_breakdown"].items()})
PY
