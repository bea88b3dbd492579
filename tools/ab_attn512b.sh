# in-call A/B of two builds on the d=512 attention (one-wave-per-SIMD kernel): baseline library vs current
# usage: bash tools/ab_attn512b.sh <baseline.so>
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; BASE=$R/$1; L=gpurun_out/ab_attn512b.log; : > $L
echo "== tests (UAV_ATTN512=1: new kernel at every size)" >> $L
UAV_ATTN512=1 timeout 600 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py $R/tests/test_parity_r2_gpu.py $R/tests/test_models_gpu.py -m gpu -q -k "attention or vae" 2>&1 | tail -4 >> $L
for r in 1 2; do for lib in "$BASE" ""; do
  echo "== bench_kernels attn lib=${lib:-current} round $r" >> $L
  UAV_HIP_LIB=$lib timeout 120 python $R/tools/bench_kernels.py attn 2>&1 | grep '"d": 512' >> $L
done; done
for lib in "$BASE" ""; do
  UAV_HIP_LIB=$lib timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('e2e lib=${lib:-current}', round(d['value'],4), round(d['ms_per_step'],1), 'attn512', kb['attention_d512'])" >> $L
done
cat $L
