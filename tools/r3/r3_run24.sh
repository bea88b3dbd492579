#!/bin/bash
# round 3, GPU call 24: a 32-frame clip (5 unique temporal windows per DDIM step, 11 decode chunks) with its windows + chunks on two
# HIP streams vs serial, same box, 6 DDIM steps
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_ab_windows_on_two_streams.log
: > $L
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'frames/s=%.4f'%d['value'], 'ms/clip=%.1f'%d['ms_per_step'], 'sha=',d['config']['output_sha256'][:16])
" | tee -a $L; }
B="--frames 32 --ddim-steps 6 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --digest --text-encoder standin"
for ov in 0 2; do
  timeout 200 python bench.py $B --overlap-streams $ov 2>/dev/null | line "32-frame clip, 6 DDIM steps, windows + decode chunks on $ov streams (0 = serial):"
done
