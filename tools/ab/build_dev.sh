#!/bin/bash
# Development instances that are NOT part of libuav_hip.so (trace / ablation kernels stay out of the product library):
# compiled with -DUAV_DEV_KERNELS into small side libraries the tools/trace_*.py scripts load with ctypes.  Run here (hipcc cross-compiles);
# the .so files travel to the GPU box with the snapshot (git-ignored).
#   tools/ab/libuav_xattn_dev.so   csrc/xattn_fused.hip + the s_memtime-stamped instance (uav_dev_xattn_sublayer_trace)
R=$(cd "$(dirname "$0")/../.." && pwd)
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -DUAV_DEV_KERNELS -shared \
    -o $R/tools/ab/libuav_xattn_dev.so $R/upscale-a-video_amd/csrc/xattn_fused.hip
echo built $R/tools/ab/libuav_xattn_dev.so
