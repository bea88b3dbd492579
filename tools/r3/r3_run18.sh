#!/bin/bash
# round 3, GPU call 18: the 64x64 full-width parity cases with the 256x256 kernel forced (UAV_CONV_TILE=256), so that the LayerNorm fold
# — which needs that kernel and is therefore inactive at 64x64 — takes part in the 30-step curve; with and without the fold
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0; do
  rm -f gpurun_out/parity.jsonl
  UAV_CONV_TILE=256 UAV_LN_FOLD=$v timeout 600 python -m pytest tests/test_parity_r3_gpu.py -q -m gpu -k "fp32_stream and (30_step or stream_modes)" 2>&1 | tail -2
  grep -E "r3_(unet_full_t8_64_fp32|pipe_full30_64_fp32)" gpurun_out/parity.jsonl | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln)
    if 'steps' in d: print('ln_fold=$v 30-step curve', [round(x*1e4,3) for x in d['engine_fp32_draws_vs_reference_fp32']], 'image', d['image_rel_l2_unsaturated_vs_reference_fp32'])
    else: print('ln_fold=$v forward', d['rel_l2_vs_reference_fp32'], d['rel_l2_vs_reference_fp32_with_fp32_output'])"
done | tee gpurun_out/r3_parity_ln_fold_forced_big_tile.log
