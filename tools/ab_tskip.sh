# in-call A/B: out-of-clip temporal taps skipped per tile (current library) vs multiplied with zeros (baseline library)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; BASE=$R/tools/ab/libuav_base.so; L=gpurun_out/ab_tskip.log; : > $L
for lib in "$BASE" ""; do echo "== digest lib=${lib:-current}" >> $L; UAV_HIP_LIB=$lib timeout 120 python $R/tools/conv_digest.py 2>&1 | grep -v amdgpu.ids >> $L; done
echo "== tests" >> $L
timeout 600 python -m pytest $R/tests -m gpu -q 2>&1 | tail -3 >> $L
for lib in "$BASE" ""; do echo "== bench_kernels lib=${lib:-current}" >> $L
  UAV_HIP_LIB=$lib timeout 120 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | grep '"t[35] ' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-34s %7.3f ms %6.0f TF (all taps counted)' % (d['case'], d['ms'], d['tflops']))" >> $L
done
for r in 1 2; do for lib in "$BASE" ""; do
  UAV_HIP_LIB=$lib timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']; print('e2e lib=${lib:-current}', round(d['value'],4), round(d['ms_per_step'],1), 'conv ms', kb['conv_gemm']['ms'])" >> $L
done; done
cat $L
