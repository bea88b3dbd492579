#!/bin/bash
# round 3, GPU call 3: 30-step curve (both stream modes) vs the reference fixture, unit tests after the raw-copy change, bench f32
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_r3_gpu.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r3_tests_parity_r3.log
tail -6 gpurun_out/r3_tests_parity_r3.log
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_parity_r3_gpu.py -x 2>&1 | tail -15 > gpurun_out/r3_tests_all.log
tail -4 gpurun_out/r3_tests_all.log
timeout 600 python bench.py --steps 2 --no-cpu-baseline --unet-stream f32 > gpurun_out/r3_bench_f32_b.json 2> gpurun_out/r3_bench_f32_b.err
timeout 600 python bench.py --steps 2 --no-cpu-baseline --unet-stream f16 > gpurun_out/r3_bench_f16_b.json 2> gpurun_out/r3_bench_f16_b.err
python - <<'PY'
import json
for m in ("f16_b", "f32_b"):
    try:
        d = json.load(open(f"gpurun_out/r3_bench_{m}.json"))
        print(m, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["kernel_time_ms_per_step"])
        print({k: v["ms"] for k, v in d["kernel_breakdown"].items()})
    except Exception as e:
        print(m, "failed", e); print(open(f"gpurun_out/r3_bench_{m}.err").read()[-1500:])
PY
grep r3_ gpurun_out/parity.jsonl
