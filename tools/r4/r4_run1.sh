#!/bin/bash
# round 4, GPU call 1: (a) the never-run 8-phase k-loop candidate (digests, micro-benchmarks, clip A/B) under a timeout,
# (b) headline-shape end-to-end parity of configs[1] / configs[2] vs the GPU oracle, (c) the precision-knob table at that shape.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 700 bash tools/next/ab_8phase.sh > gpurun_out/r4_run1_ab8.out 2>&1
echo "ab_8phase rc=$?" >> gpurun_out/r4_run1_ab8.out
UAV_R4_SAVE_ORACLE=/tmp/r4_oracle.pt timeout 900 python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4_run1_parity_tests.log
cp gpurun_out/parity.jsonl gpurun_out/r4_run1_parity.jsonl
timeout 600 python tools/r4/parity_variants.py /tmp/r4_oracle.pt > gpurun_out/r4_run1_parity_variants.jsonl 2> gpurun_out/r4_run1_parity_variants.err
cat gpurun_out/ab_conv_8phase.log
cat gpurun_out/r4_run1_parity_tests.log
python - <<'PY'
import json
for l in open('gpurun_out/r4_run1_parity.jsonl'):
    d = json.loads(l)
    c = d.pop('latents_rel_l2_per_step', None)
    if c: d['curve_1_5_10_20_24_30'] = [c[0], c[4], c[9], c[19], c[23], c[29]]
    print(d)
PY
cat gpurun_out/r4_run1_parity_variants.jsonl; tail -3 gpurun_out/r4_run1_parity_variants.err
