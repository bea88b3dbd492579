# in-call A/B: 128x128 kernel (2 workgroups per CU) vs 256x256 kernel (1 per CU) per shape
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
for v in 128 256 128 256; do
  echo "== UAV_CONV_TILE=$v"; UAV_CONV_TILE=$v timeout 100 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))"
done 2>&1 | tee gpurun_out/ab_tile.log
