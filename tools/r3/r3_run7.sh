#!/bin/bash
# round 3, GPU call 7: temporal attention v2 (loads issued together) A/B inside one box, kernel tests
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -m gpu -x 2>&1 | tail -3
UAV_TATTN=1 timeout 200 python tools/bench_kernels.py tattn 2>/dev/null | sed 's/^/v1 /'
UAV_TATTN=2 timeout 200 python tools/bench_kernels.py tattn 2>/dev/null | sed 's/^/v2 /'
for i in 1 2; do
  for v in 1 2; do
    UAV_TATTN=$v timeout 200 python bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('tattn v$v', round(d['value'],4), round(d['ms_per_step'],1), 'tattn', kb['temporal_attention']['ms'], kb['temporal_attention']['GBps'])"
  done
done | tee gpurun_out/r3_ab_temporal_attention_v2.log
