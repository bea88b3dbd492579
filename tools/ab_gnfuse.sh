# in-call A/B: GroupNorm statistics fused into the conv epilogue (UAV_FUSE_GN_STATS=1) vs the stand-alone pass (=0),
# and the walk order of the stand-alone GroupNorm passes (UAV_GN_ORDER bit 0: statistics backwards, bit 1: apply backwards)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; L=gpurun_out/ab_gnfuse.log; : > $L
echo "== tests" >> $L
timeout 600 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py -m gpu -q -k "conv or linear or geglu or fusions or f32_stream or groupnorm or determinism" 2>&1 | tail -6 >> $L
echo "== micro" >> $L
timeout 300 python $R/tools/bench_gnfuse.py 2>&1 | grep -v amdgpu.ids >> $L
run() {
  UAV_FUSE_GN_STATS=$1 UAV_GN_ORDER=$2 timeout 300 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('e2e FUSE=$1 ORDER=$2', round(d['value'],4), round(d['ms_per_step'],1), 'conv TF', round(d['roofline']['achieved'],1), 'conv ms', round(kb['conv_gemm']['ms'],1), 'gn_stats ms', round(kb['groupnorm_stats']['ms'],1), 'n', kb['groupnorm_stats']['launches'], 'fused fin ms', round(kb.get('groupnorm_finalize_fused',{}).get('ms',0),1), kb.get('groupnorm_finalize_fused',{}).get('launches',0), 'apply ms', round(kb['groupnorm_apply']['ms'],1))" >> $L
}
run 0 0; run 1 0; run 0 1; run 0 2; run 0 3; run 0 0; run 1 0
cat $L
