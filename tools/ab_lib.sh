# same-box A/B of two builds of libuav_hip.so on the end-to-end bench (box-to-box variance makes cross-call numbers useless)
# usage: bash tools/ab_lib.sh <baseline.so> [bench args]
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; BASE=$1; shift
for i in 1 2; do
  for lib in "$R/$BASE" ""; do
    UAV_HIP_LIB=$lib timeout 150 python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-current}', round(d['value'],4), round(d['ms_per_step'],1))"
  done
done | tee gpurun_out/ab_lib.log
