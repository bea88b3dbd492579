# in-call A/B of the GroupNorm statistics kernel variants (UAV_GN_VAR)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_gn.log; : > $L
for v in 1 2 3; do echo "== tests GN_VAR=$v" >> $L
  UAV_GN_VAR=$v timeout 300 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py -m gpu -q -k "groupnorm" 2>&1 | tail -2 >> $L; done
for r in 1 2; do for v in 0 1 2 3; do
  echo "== bench_kernels GN_VAR=$v round $r" >> $L
  UAV_GN_VAR=$v timeout 120 python $R/tools/bench_kernels.py norm 2>&1 | grep gn_stats >> $L
done; done
for r in 1 2; do for v in 0 1 2 3; do
  UAV_GN_VAR=$v timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e GN_VAR=$v', round(d['value'],4), round(d['ms_per_step'],1), 'gn_stats', d['kernel_breakdown']['groupnorm_stats'], 'kernel_ms', round(d['kernel_time_ms_per_step'],1))" >> $L
done; done
cat $L
