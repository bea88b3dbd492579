#!/bin/bash
# round 3, GPU call 27: is the slow "linear + fp32 residual -> fp32 + GroupNorm statistics" case of run 26 (640 instances of 640
# rows) a property of the product's shapes (2 instances of 204800 rows)?  Same epilogue at the product geometry, 256- and 128-tile.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_ab_1x1_gn_fp32_epilogue_geometry.log
: > $L
for t in 0 128; do
  echo "UAV_CONV_TILE=$t" | tee -a $L
  UAV_CONV_TILE=$t timeout 100 python tools/bench_epilogue.py "1x1" "res32->f32+gn" 2>&1 | grep '^{' | tee -a $L
done
