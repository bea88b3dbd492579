# same-box A/B of two builds of libuav_hip.so on the end-to-end bench (box-to-box variance makes cross-call numbers useless)
# usage: bash tools/ab_lib.sh <baseline.so> [pytest -k expression]
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; BASE=$1; KEXPR=${2:-"attention or unet_forward"}
timeout 300 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_models_gpu.py $R/tests/test_fullsize_gpu.py -m gpu -x -q -n 2 -k "$KEXPR" 2>&1 | tail -2
for i in 1 2; do
  for lib in "$R/$BASE" ""; do
    UAV_HIP_LIB=$lib timeout 150 python $R/bench.py --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print('${lib:-current}', round(d['value'],4), round(d['ms_per_step'],1), 'tattn', kb['temporal_attention']['ms'], 'attn512', kb['attention_d512']['ms'], 'attn64', kb['attention_d64']['ms'], 'attn128', kb['attention_d128']['ms'])"
  done
done | tee gpurun_out/ab_lib.log
