#!/bin/bash
# round 3, GPU call 17: LayerNorm folded into the consuming projection: kernel test, parity, same-box A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_parity_r3_gpu.py -q -m gpu 2>&1 | tail -12
grep -E "r3_(unet|headline_unet|pipe_full30_64_fp32)" gpurun_out/parity.jsonl | cut -c1-330
for i in 1 2; do
  for v in 0 1; do
    UAV_LN_FOLD=$v timeout 200 python bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('ln_fold=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv', kb['conv_gemm']['ms'], kb['conv_gemm']['tflops'], 'layernorm', kb.get('layernorm',{}).get('ms'))"
  done
done | tee gpurun_out/r3_ab_layernorm_folded_into_projection.log
