# NOTE: UAV_CONV_HALO only exists in commit 45fd3db (the haloed-X kernel was removed after this A/B); kept as the record of how profiles/r02_ab_conv_haloed_x_image_run31.log was produced
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_halo2.log; : > $L
for r in 1 2; do for v in 0 1; do
  echo "== bench_kernels HALO=$v round $r" >> $L
  UAV_CONV_HALO=$v timeout 120 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | grep "3x3" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))" >> $L
done; done
for r in 1 2; do for v in 0 1; do
  UAV_CONV_HALO=$v timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e HALO=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv TF', round(d['roofline']['achieved'],1), 'conv ms', d['kernel_breakdown']['conv_gemm']['ms'])" >> $L
done; done
cat $L
