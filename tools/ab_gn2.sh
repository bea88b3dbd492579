# same-box A/B: per-group GroupNorm partials + cheap finalize (current) vs per-channel partials (tools/ab/libuav_base.so)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_gn2.log; : > $L
timeout 400 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_models_gpu.py $R/tests/test_fullsize_gpu.py -m gpu -q -k "groupnorm or unet_forward or vae or resnet or f32_stream" 2>&1 | tail -3 >> $L
for i in 1 2; do for lib in "$R/tools/ab/libuav_base.so" ""; do
  UAV_HIP_LIB=$lib timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('${lib:-current}', round(d['value'],4), round(d['ms_per_step'],1), 'gn_stats', kb['groupnorm_stats']['ms'], 'gn_apply', kb['groupnorm_apply']['ms'], 'kernel_ms', round(d['kernel_time_ms_per_step'],1))" >> $L
done; done
cat $L
