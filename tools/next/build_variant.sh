#!/bin/bash
# Builds tools/ab/libuav_hip_8phase.so = the product library with csrc/conv_gemm.hip + conv_gemm_8phase.patch (the 8-phase
# k-loop candidate, conv_gemm256p_kernel, selected at run time by UAV_CONV_DMAV=8).  The product sources are not touched.
# usage: bash tools/next/build_variant.sh     (needs uav/build/*.o of a fresh product build: python __graft_entry__.py)
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
W=$(mktemp -d)
cp "$R/upscale-a-video_amd/csrc/conv_gemm.hip" "$W/conv_gemm.hip"
patch -s "$W/conv_gemm.hip" "$R/tools/next/conv_gemm_8phase.patch"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I"$R/upscale-a-video_amd/csrc" -I"$R/include" \
    -c "$W/conv_gemm.hip" -o "$W/conv_gemm.o" -Rpass-analysis=kernel-resource-usage 2> "$W/res.txt"
grep -A5 "conv_gemm256p_kernel" "$W/res.txt" | grep -E "Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: //'
B="$R/upscale-a-video_amd/uav/build"
mkdir -p "$R/tools/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/ab/libuav_hip_8phase.so" "$W/conv_gemm.o" \
    "$B/attention.o" "$B/colorfix.o" "$B/conv_gemm_f32.o" "$B/elementwise.o" "$B/norm.o" "$B/raft.o" "$B/temporal_attn.o"
rm -rf "$W"
echo "built $R/tools/ab/libuav_hip_8phase.so"
