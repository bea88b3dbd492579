# in-call A/B of two library builds on the conv microbench + end-to-end (box-to-box variance is large: same-box only)
# usage: bash tools/ab_conv.sh <baseline.so>
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; BASE=$R/$1
timeout 250 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py -m gpu -x -q -n 2 -k "conv or linear or geglu or fusions" 2>&1 | tail -2
for lib in "$BASE" "" "$BASE" ""; do
  echo "== ${lib:-current}"; UAV_HIP_LIB=$lib timeout 100 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))"
done 2>&1 | tee gpurun_out/ab_conv_epilogue.log
for i in 1 2; do for lib in "$BASE" ""; do
    UAV_HIP_LIB=$lib timeout 150 python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-current}', round(d['value'],4), round(d['ms_per_step'],1))"
done; done | tee gpurun_out/ab_lib.log
