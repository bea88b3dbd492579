#!/bin/bash
# round 3, GPU call 26: (a) the short-K linears of the fp32 stream on the 128x128-tile kernel (2-3 workgroups per CU: one tile's
# epilogue beside another's main loop) vs the 256x256-tile kernel (one workgroup per CU), same box; (b) the stream tests on the final tree
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_ab_short_k_linears_tile128_vs_256.log
: > $L
for t in 0 128; do
  echo "UAV_CONV_TILE=$t" | tee -a $L
  UAV_CONV_TILE=$t timeout 120 python tools/bench_epilogue.py "linear" 2>&1 | grep '^{' | tee -a $L
done
timeout 120 python -m pytest tests/test_models_gpu.py -q -m gpu -k "concurrent_streams or long_clip" 2>&1 | tail -2 | tee -a $L
