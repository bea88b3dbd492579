# NOTE: UAV_CONV_STAGGER was a one-run experiment (no effect, profiles/r02_ab_conv_first_wave_stagger_run28.log); the switch is not in the tree
# in-call A/B: first-wave stagger of the short-K conv launches (UAV_CONV_STAGGER = 1024-cycle sleep units per k-step)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_stagger.log; : > $L
for v in 0 1 2 3 5; do echo "== bench_kernels conv STAGGER=$v" >> $L
  UAV_CONV_STAGGER=$v timeout 120 python $R/tools/bench_kernels.py conv 2>&1 | grep linear | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))" >> $L
done
for r in 1 2; do for v in 0 2 3; do
  UAV_CONV_STAGGER=$v timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e STAGGER=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv TF', round(d['roofline']['achieved'],1), 'conv ms', d['kernel_breakdown']['conv_gemm']['ms'])" >> $L
done; done
cat $L
