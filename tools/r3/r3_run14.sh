#!/bin/bash
# round 3, GPU call 14: short-key attention kernel (text cross-attention, all K / V tiles requested up front): tests + same-box A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_clip_text_gpu.py -q -m gpu -x 2>&1 | tail -3
UAV_ATTN_SHORT=0 timeout 200 python tools/bench_kernels.py attn 2>/dev/null | sed 's/^/old /'
UAV_ATTN_SHORT=1 timeout 200 python tools/bench_kernels.py attn 2>/dev/null | sed 's/^/new /'
for i in 1 2; do
  for v in 0 1; do
    UAV_ATTN_SHORT=$v timeout 200 python bench.py --no-cpu-baseline --steps 1 --digest 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('attn_short=$v', round(d['value'],4), round(d['ms_per_step'],1), 'd64', kb['attention_d64']['ms'], kb['attention_d64']['GBps'], 'd128', kb['attention_d128']['ms'], d['config']['output_sha256'][:16])"
  done
done | tee gpurun_out/r3_ab_attention_short_keys.log
