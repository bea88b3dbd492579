#!/bin/bash
# round 3, GPU call 8: GroupNorm statistics of concatenated inputs from the two producers' partials: test + same-box A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
  for v in 0 1; do
    UAV_GN_TWO_SOURCE=$v timeout 200 python bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('two_source=$v', round(d['value'],4), round(d['ms_per_step'],1), 'gn_stats', kb['groupnorm_stats']['ms'], kb['groupnorm_stats']['launches'], 'finalize', kb['groupnorm_finalize_fused']['ms'], kb['groupnorm_finalize_fused']['launches'])"
  done
done | tee gpurun_out/r3_ab_groupnorm_two_source_partials.log
