# in-call A/B: skip tensors of the CFG-shared head read batch-broadcast (UAV_BROADCAST_SKIPS=1, default) vs duplicated with cat (=0)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_bcast.log; : > $L
timeout 300 python -m pytest $R/tests -m gpu -q 2>&1 | tail -3 >> $L
for v in 0 1; do
  UAV_BROADCAST_SKIPS=$v timeout 120 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']; print('e2e BCAST=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv ms', kb['conv_gemm']['ms'], 'gn', kb['groupnorm_stats']['ms'], kb['groupnorm_apply']['ms'])" >> $L
done
cat $L
