#!/bin/bash
# round 3, GPU call 5: new defaults (fp32 stream + hi/lo shortcut operands + epilogue order): whole GPU suite, smoke, parity
# report, default bench, fp16-stream bench, per-shape table
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r3_tests_all.log
tail -6 gpurun_out/r3_tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 2 --no-cpu-baseline > gpurun_out/r3_bench_default_c.json 2> gpurun_out/r3_bench_default_c.err
timeout 600 python bench.py --steps 2 --no-cpu-baseline --unet-stream f16 > gpurun_out/r3_bench_f16_c.json 2> gpurun_out/r3_bench_f16_c.err
UAV_BENCH_DETAIL=1 timeout 600 python bench.py --steps 1 --no-cpu-baseline > gpurun_out/r3_detail_default_c.json 2> gpurun_out/r3_detail_default_c.txt
UAV_SHORTCUT_HILO=0 timeout 600 python bench.py --steps 2 --no-cpu-baseline > gpurun_out/r3_bench_nohilo_c.json 2> gpurun_out/r3_bench_nohilo_c.err
python - <<'PY'
import json
for m in ("default_c", "f16_c", "nohilo_c"):
    try:
        d = json.load(open(f"gpurun_out/r3_bench_{m}.json"))
        print(m, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["kernel_time_ms_per_step"])
        print({k: v["ms"] for k, v in d["kernel_breakdown"].items()})
    except Exception as e:
        print(m, "failed", e); print(open(f"gpurun_out/r3_bench_{m}.err").read()[-1500:])
PY
grep -E "r3_|pipe_c1_full|unet_full|vae3d_full" gpurun_out/parity.jsonl
