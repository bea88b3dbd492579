# in-call A/B of conv kernel switches (box-to-box variance is up to 20 %, only same-box comparisons count)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 200 python -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -n 2 -k "conv or linear or geglu" 2>&1 | tail -2
for v in "UAV_CONV_PREFETCH=0" "UAV_CONV_PREFETCH=1" "UAV_CONV_PREFETCH=0" "UAV_CONV_PREFETCH=1"; do
  echo "== $v"; env $v timeout 100 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'linear' in d['case']: print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))"
done 2>&1 | tee gpurun_out/ab_conv_prefetch.log
for v in "UAV_CONV_PREFETCH=0" "UAV_CONV_PREFETCH=1"; do env $v timeout 120 python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"; done | tee -a gpurun_out/ab_conv_prefetch.log
