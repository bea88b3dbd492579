# in-call A/B: d=512 attention, pair-split kernel (UAV_ATTN512=0) vs one-wave-per-SIMD kernel with the O^T tile in the accumulator file (=1)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_attn512.log; : > $L
for v in 0 1; do echo "== tests UAV_ATTN512=$v" >> $L
  UAV_ATTN512=$v timeout 600 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py $R/tests/test_parity_r2_gpu.py $R/tests/test_models_gpu.py -m gpu -q -k "attention or vae" 2>&1 | tail -4 >> $L; done
for r in 1 2; do for v in 0 1; do
  echo "== bench_kernels attn UAV_ATTN512=$v round $r" >> $L
  UAV_ATTN512=$v timeout 120 python $R/tools/bench_kernels.py attn 2>&1 | grep '"d": 512' >> $L
done; done
for r in 1 2; do for v in 0 1; do
  UAV_ATTN512=$v timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('e2e UAV_ATTN512=$v', round(d['value'],4), round(d['ms_per_step'],1), 'attn512', kb['attention_d512'])" >> $L
done; done
cat $L
