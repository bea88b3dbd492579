#!/bin/bash
# round 3, GPU call 1: new parity tests (stream modes, headline shape vs the GPU oracle), whole GPU suite, bench in both UNet stream modes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_r3_gpu.py -q -m gpu -k "not 30_step" 2>&1 | tail -25 > gpurun_out/r3_tests_parity_r3.log
cat gpurun_out/r3_tests_parity_r3.log | tail -8
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_parity_r3_gpu.py 2>&1 | tail -15 > gpurun_out/r3_tests_all.log
tail -5 gpurun_out/r3_tests_all.log
timeout 600 python bench.py --steps 2 --no-cpu-baseline --unet-stream f16 > gpurun_out/r3_bench_f16.json 2> gpurun_out/r3_bench_f16.err
timeout 600 python bench.py --steps 2 --no-cpu-baseline --unet-stream f32 > gpurun_out/r3_bench_f32.json 2> gpurun_out/r3_bench_f32.err
python - <<'PY'
import json
for m in ("f16", "f32"):
    try:
        d = json.load(open(f"gpurun_out/r3_bench_{m}.json"))
        print(m, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["kernel_time_ms_per_step"])
        print({k: v["ms"] for k, v in d["kernel_breakdown"].items()})
    except Exception as e:
        print(m, "failed", e); print(open(f"gpurun_out/r3_bench_{m}.err").read()[-1500:])
PY
cat gpurun_out/parity.jsonl
