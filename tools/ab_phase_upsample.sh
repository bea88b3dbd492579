# in-call A/B: upsampling convs as four 2x2 sub-pixel phase convs (UAV_PHASE_UPSAMPLE=1, default) vs the fused-gather 3x3 form (=0)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_phase_upsample.log; : > $L
echo "== tests" >> $L
timeout 600 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_models_gpu.py $R/tests/test_fullsize_gpu.py $R/tests/test_parity_r2_gpu.py -m gpu -q -k "upsample or conv or unet or vae or pipeline or full" 2>&1 | tail -4 >> $L
for r in 1 2; do for v in 0 1; do
  UAV_PHASE_UPSAMPLE=$v timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']; print('e2e PHASE=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv TF', round(d['roofline']['achieved'],1), 'conv ms', kb['conv_gemm']['ms'], 'launches', kb['conv_gemm']['launches'], 'gn_stats', kb['groupnorm_stats']['ms'])" >> $L
done; done
cat $L
