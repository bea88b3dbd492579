#!/bin/bash
# GPU A/B of the 8-phase k-loop candidate (run on the GPU box through gpurun; build the variant first, it travels with the tree):
#   1. bit identity: output digests of seven conv shapes, UAV_CONV_DMAV=8 vs 1 (same K order per accumulator -> same bits expected)
#   2. micro-benchmarks with the round-3 epilogues (tools/bench_epilogue.py), 30-iteration timings
#   3. the headline clip, both kernels, same box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp UAV_HIP_LIB=$PWD/tools/ab/libuav_hip_variant.so UAV_CONV_TILE=256
L=gpurun_out/ab_conv_8phase.log
: > $L
for v in 1 8; do echo "digests UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v timeout 120 python tools/conv_digest.py 2>&1 | tail -1 | tee -a $L; done
unset UAV_CONV_TILE
for v in 1 8; do echo "micro-benchmarks UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v UAV_EPI_ITERS_X=5 timeout 200 python tools/bench_epilogue.py 2>&1 | grep '^{' | tee -a $L; done
for v in 1 8 1 8; do
  UAV_CONV_DMAV=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --digest 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline',{})
print('UAV_CONV_DMAV=$v frames/s=%.4f ms/clip=%.1f conv TFLOP/s=%.1f sha=%s' % (d['value'], d['ms_per_step'], r.get('achieved',0), d['config']['output_sha256'][:16]))" | tee -a $L
done
