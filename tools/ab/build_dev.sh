#!/bin/bash
# Development kernels are NOT part of libuav_hip.so (trace / ablation / legacy instances stay out of the product library).
# This script builds, with -DUAV_DEV_KERNELS:
#   tools/ab/libuav_xattn_dev.so   csrc/xattn_fused.hip + its s_memtime-stamped instance (tools/trace_xattn.py loads it with ctypes)
#   tools/ab/libuav_hip_dev.so     (with `all`) the WHOLE library incl. csrc/conv_gemm_dev.hip: round-1 conv family + ablation builds, the
#                                  round 2-3 loop and its LayerNorm-fold instances, the short-K kernel, the stamped four-wave conv instance,
#                                  attn512_kernel / attn_kernel<512>.  Same C ABI: UAV_HIP_LIB=tools/ab/libuav_hip_dev.so selects it
#                                  (tools/trace_w4.py, tools/bench_shortk.py, UAV_LN_FOLD=1, UAV_CONV_DBG / _PERSIST / _DMAV / _SK, UAV_ATTN512=0).
# Run here (hipcc cross-compiles); the .so files travel to the GPU box with the snapshot (git-ignored).
R=$(cd "$(dirname "$0")/../.." && pwd)
set -e
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Wno-unused-function -DUAV_DEV_KERNELS"
/opt/rocm/bin/hipcc $F -I$R/include -shared -o $R/tools/ab/libuav_xattn_dev.so $R/upscale-a-video_amd/csrc/xattn_fused.hip
echo built $R/tools/ab/libuav_xattn_dev.so
if [ "$1" = "all" ]; then
  mkdir -p /tmp/uav_dev_obj
  pids=""
  for f in $R/upscale-a-video_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc $F -c $f -o /tmp/uav_dev_obj/$(basename ${f%.hip}).o & pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libuav_hip_dev.so /tmp/uav_dev_obj/*.o
  echo built $R/tools/ab/libuav_hip_dev.so
fi
