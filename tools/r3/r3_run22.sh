#!/bin/bash
# round 3, GPU call 22: one clip's independent units on concurrent HIP streams (uav/streams.py, pipeline.overlap_streams):
# same bits as the serial order (test + output digests), and a same-box A/B of the headline clip: serial / 2 streams / 3 streams
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_ab_units_on_concurrent_streams.log
: > $L
timeout 600 python -m pytest tests/test_models_gpu.py -q -m gpu -k "concurrent_streams or end_to_end or cfg_shared" 2>&1 | tail -4 | tee -a $L
for ov in 0 2 0 2 3; do
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --digest --overlap-streams $ov 2> gpurun_out/ov_err_$ov.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d.get('roofline',{})
print('overlap_streams=$ov', 'frames/s=%.4f'%d['value'], 'ms/clip=%.1f'%d['ms_per_step'], 'sha=',d['config']['output_sha256'][:16], 'conv TFLOP/s (serial step)=%.1f'%r.get('achieved',0), 'kernel ms/step=%.0f'%d.get('kernel_time_ms_per_step',0))
" | tee -a $L
  tail -3 gpurun_out/ov_err_$ov.txt | grep -i -E "error|Traceback" | tee -a $L
done
