mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
UAV_CONV_PERSIST=1 timeout 120 python -m pytest $R/tests/test_fullsize_gpu.py $R/tests/test_kernels_gpu.py -m gpu -x -q -k "conv or linear or fusions" 2>&1 | tail -2
for v in 0 1 0 1; do
  echo "== UAV_CONV_PERSIST=$v"; UAV_CONV_PERSIST=$v timeout 60 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if any(k in d['case'] for k in ('linear', '512->512 @16x320', 't3 ')): print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))"
done 2>&1 | tee gpurun_out/ab_persist.log
