#!/bin/bash
# round 3, GPU call 29: EXPERIMENT (variant library tools/ab/stagger/conv_gemm.hip, product sources untouched): do the workgroups
# of a short-K linear run their main loops and their HBM-heavy epilogues in chip-wide lockstep?  First-round workgroups of
# every other CU slot sleep UAV_CONV_STAGGER x 3.9 us before their first tile; 0 = off.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp UAV_HIP_LIB=$PWD/tools/ab/libuav_hip_stagger.so UAV_EPI_ITERS_X=5
L=gpurun_out/r3_ab_conv_phase_stagger.log
: > $L
for s in 0 2 4 6 0; do
  echo "UAV_CONV_STAGGER=$s" | tee -a $L
  UAV_CONV_STAGGER=$s timeout 60 python tools/bench_epilogue.py "linear 512->512 M=409600 bias+res32->f32" "linear 512->512 M=409600 bias+res16" "linear 2048->512 M=409600 bias+res32" "3x3 512->512 @16x160x160 bias+res32" 2>&1 | grep '^{' | tee -a $L
done
