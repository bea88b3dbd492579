#!/bin/bash
# round 4, GPU call 8: the ROTATED k-step (UAV_CONV_DMAV=5 / 6; product library) against the product loop (=1): digests, conv kernel tests
# on the new loops, micro-benchmarks, clip A/B.  (The product loop itself was rebuilt: its epilogue-constant staging now selects a pointer
# first — the statistics instances lost their scratch — so its digests are compared with the committed round-4 values as well.)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_run8_ab_rotated_kstep.log; : > $L
export UAV_CONV_TILE=256
for v in 1 5 6 5 6; do echo "digests UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v timeout 120 python tools/conv_digest.py 2>&1 | tail -1 | tee -a $L; done
unset UAV_CONV_TILE
for v in 5 6; do
  echo "conv kernel tests on UAV_CONV_DMAV=$v" | tee -a $L
  UAV_CONV_DMAV=$v timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "conv or gn or linear or upsampl or shortcut" 2>&1 | tail -2 | tee -a $L
done
for v in 1 5 6; do echo "micro-benchmarks UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v UAV_EPI_ITERS_X=5 timeout 200 python tools/bench_epilogue.py 2>&1 | grep '^{' | tee -a $L; done
for v in 1 5 6 1 5 6; do
  UAV_CONV_DMAV=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --digest 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline',{})
print('UAV_CONV_DMAV=$v frames/s=%.4f ms/clip=%.1f conv TFLOP/s=%.1f sha=%s' % (d['value'], d['ms_per_step'], r.get('achieved',0), d['config']['output_sha256'][:16]))" | tee -a $L
done
