#!/bin/bash
# round 3, GPU call 4: epilogue vector-memory order (staged bias, residual loads ahead of the stores): kernel tests, same-box A/B
# of the two library builds on the conv micro-benchmarks (with the epilogue variants that matter) and end to end
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python tools/bench_epilogue.py > gpurun_out/r3_ab_epilogue_new.jsonl 2>&1
UAV_HIP_LIB=$R/tools/ab/libuav_base.so timeout 300 python tools/bench_epilogue.py > gpurun_out/r3_ab_epilogue_base.jsonl 2>&1
python - <<'PY'
import json
def load(f):
    d = {}
    for ln in open(f):
        if ln.startswith("{"):
            r = json.loads(ln); d[r["case"]] = r
    return d
a, b = load("gpurun_out/r3_ab_epilogue_base.jsonl"), load("gpurun_out/r3_ab_epilogue_new.jsonl")
for k in a:
    if k in b:
        print(f"{k:58s} base {a[k]['ms']:8.3f} ms {a[k]['tflops']:7.1f} TF/s   new {b[k]['ms']:8.3f} ms {b[k]['tflops']:7.1f} TF/s   {100 * (b[k]['ms'] / a[k]['ms'] - 1):+6.1f} %")
PY
for i in 1 2; do
  for lib in "$R/tools/ab/libuav_base.so" ""; do
    for m in f16 f32; do
      UAV_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --steps 1 --unet-stream $m 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('${lib:-current}'[-16:], '$m', round(d['value'],4), round(d['ms_per_step'],1), 'conv', kb['conv_gemm']['ms'], kb['conv_gemm']['tflops'], 'gn_apply', kb['groupnorm_apply']['ms'])"
    done
  done
done | tee gpurun_out/r3_ab_epilogue_e2e.log
