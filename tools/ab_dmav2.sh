# NOTE: UAV_CONV_DMAV=4 (5-slot ring) only exists in commit 70468d9; the loops below now run the shipped variant only
# in-call A/B: DMAV=1 (interleaved DMA, 2 stages) vs DMAV=4 (5-slot ring, counted vmcnt)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_dmav2.log; : > $L
for v in 1; do echo "== digest DMAV=$v" >> $L; UAV_CONV_DMAV=$v timeout 120 python $R/tools/conv_digest.py 2>&1 | grep -v amdgpu.ids >> $L; done
echo "== tests DMAV=4" >> $L
UAV_CONV_DMAV=4 timeout 300 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py -m gpu -q -k "conv or linear or geglu or fusions or f32_stream" 2>&1 | tail -2 >> $L
for r in 1 2; do for v in 1; do
  echo "== bench_kernels DMAV=$v round $r" >> $L
  UAV_CONV_DMAV=$v timeout 120 python $R/tools/bench_kernels.py conv 2>&1 | grep conv_gemm | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  %-34s %7.3f ms %6.0f TF' % (d['case'], d['ms'], d['tflops']))" >> $L
done; done
for r in 1 2; do for v in 1; do
  UAV_CONV_DMAV=$v timeout 200 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e DMAV=$v', round(d['value'],4), round(d['ms_per_step'],1), 'conv TF', round(d['roofline']['achieved'],1))" >> $L
done; done
cat $L
