#!/bin/bash
# Builds a VARIANT of libuav_hip.so beside the product library: the conv kernel source the candidate patches of this directory were
# written against (csrc/conv_gemm.hip of commit f5d0ffa, the start of round 4 — `git show`, so a git checkout is needed) + one patch
# (conv_gemm_8phase.patch -> UAV_CONV_DMAV=8, conv_gemm_early_release.patch -> UAV_CONV_DMAV=4; both also keep =1, the round-3 loop),
# linked with the CURRENT objects of the other kernels.  The product sources are not touched; load the result through UAV_HIP_LIB.
# Needs uav/build/*.o of a fresh product build (python __graft_entry__.py).
# usage: bash tools/r4/build_variant.sh [patch (default conv_gemm_8phase.patch)] [output (default tools/ab/libuav_hip_variant.so)]
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
P="${1:-conv_gemm_8phase.patch}"; OUT="${2:-$R/tools/ab/libuav_hip_variant.so}"
W=$(mktemp -d)
git -C "$R" show f5d0ffa:upscale-a-video_amd/csrc/conv_gemm.hip > "$W/conv_gemm.hip"
patch -s "$W/conv_gemm.hip" "$R/tools/r4/$P"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I"$R/upscale-a-video_amd/csrc" -I"$R/include" \
    -c "$W/conv_gemm.hip" -o "$W/conv_gemm.o" -Rpass-analysis=kernel-resource-usage 2> "$W/res.txt"
grep -E "Function Name|VGPRs:|ScratchSize" "$W/res.txt" | sed 's/.*remark: //' | paste - - - | grep -E "256p|ILi4E" || true
B="$R/upscale-a-video_amd/uav/build"
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$W/conv_gemm.o" \
    "$B/attention.o" "$B/colorfix.o" "$B/conv_gemm_f32.o" "$B/elementwise.o" "$B/norm.o" "$B/raft.o" "$B/temporal_attn.o"
rm -rf "$W"
echo "built $OUT"
