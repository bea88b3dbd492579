# in-call A/B of an environment switch of libuav_hip.so: bash tools/ab_env.sh VAR "case-substring|case-substring"
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; VAR=$1; PAT=$2
timeout 250 python -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -n 2 -k "conv" 2>&1 | tail -1
for v in 0 1 0 1; do
  echo "== $VAR=$v"; env $VAR=$v timeout 100 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | python -c "
import sys, json, re
for l in sys.stdin:
    d = json.loads(l)
    if re.search(r'$PAT', d['case']): print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))"
done 2>&1 | tee gpurun_out/ab_env_$VAR.log
for i in 1 2; do for v in 0 1; do
    env $VAR=$v timeout 150 python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],4), round(d['ms_per_step'],1))"
done; done | tee -a gpurun_out/ab_env_$VAR.log
