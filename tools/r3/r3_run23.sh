#!/bin/bash
# round 3, GPU call 23: (a) the guidance-branch units on two streams give the same bits as the SAME decomposition issued serially
# (--shard-windows --shard-cfg on one rank); (b) same-box A/B of the headline clip with only the decode chunks on two streams
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_ab_decode_chunks_on_two_streams.log
: > $L
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'frames/s=%.4f'%d['value'], 'ms/clip=%.1f'%d['ms_per_step'], 'sha=',d['config']['output_sha256'][:16])
" | tee -a $L; }
C="--steps 1 --warmup 0 --ddim-steps 6 --no-cpu-baseline --no-kernel-events --digest --text-encoder standin"
timeout 200 python bench.py $C --shard-windows --shard-cfg 2>/dev/null | line "branch units, serial (shard_cfg order, 1 rank), 6 DDIM steps:"
timeout 200 python bench.py $C --overlap-streams 2 --overlap-split-cfg 2>/dev/null | line "branch units on 2 streams, 6 DDIM steps:           "
B="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --digest"
for ov in 2 0 2; do
  timeout 300 python bench.py $B --overlap-streams $ov 2>/dev/null | line "headline clip, decode chunks on $ov streams (0 = serial):"
done
