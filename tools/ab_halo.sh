# NOTE: UAV_CONV_HALO only exists in commit 45fd3db (the haloed-X kernel was removed after this A/B); kept as the record of how profiles/r02_ab_conv_haloed_x_image_run31.log was produced
# in-call A/B: haloed X image conv kernel (UAV_CONV_HALO=1) vs the shipped one — bit-identity digests, tests, micro-bench, e2e
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_halo.log; : > $L
for v in 0 1; do echo "== digest HALO=$v (UAV_CONV_TILE=256)" >> $L; UAV_CONV_TILE=256 UAV_CONV_HALO=$v timeout 120 python $R/tools/conv_digest.py 2>&1 | grep -v amdgpu.ids >> $L; done
for v in 0 1; do echo "== digest HALO=$v (default tiles)" >> $L; UAV_CONV_HALO=$v timeout 120 python $R/tools/conv_digest.py 2>&1 | grep -v amdgpu.ids >> $L; done
echo "== tests HALO=1 TILE=256" >> $L
UAV_CONV_TILE=256 UAV_CONV_HALO=1 timeout 300 python -m pytest $R/tests/test_kernels_gpu.py -m gpu -q -k "conv or linear or geglu or fusions or f32_stream or groupnorm" 2>&1 | tail -3 >> $L
echo "== tests HALO=1" >> $L
UAV_CONV_HALO=1 timeout 300 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py $R/tests/test_models_gpu.py -m gpu -q -k "conv or fusions or f32_stream or groupnorm or unet_forward or vae or determinism or full_model" 2>&1 | tail -3 >> $L
if [ "$1" != "quick" ]; then
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
fi
cat $L
