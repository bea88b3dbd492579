#!/bin/bash
# round 3, GPU call 12: shortcut conv folded into conv2 (a2_center_tap) + fp16 branch tensor: tests, parity, same-box A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_parity_r3_gpu.py tests/test_video_io_gpu.py -q -m gpu 2>&1 | tail -8
grep -E "r3_" gpurun_out/parity.jsonl | cut -c1-420
for i in 1 2; do
  for v in 0 1; do
    UAV_FUSE_SHORTCUT=$v timeout 200 python bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('fuse_shortcut=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv', kb['conv_gemm']['ms'], kb['conv_gemm']['launches'], kb['conv_gemm']['tflops'], 'gn_apply', kb['groupnorm_apply']['ms'])"
  done
done | tee gpurun_out/r3_ab_shortcut_folded_into_conv2.log
UAV_BRANCH_F32=1 timeout 200 python bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('branch_f32=1', round(d['value'],4), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3_ab_shortcut_folded_into_conv2.log
