#!/bin/bash
# round 3, GPU call 30: VOID (UAV_CONV_PERSIST is shadowed by the default UAV_CONV_DMAV=1: both legs ran conv_gemm256i_kernel<1>; needs UAV_CONV_DMAV=0)
# first DMA stage of the NEXT tile before the epilogue of the current one; UAV_CONV_PERSIST=1) on the short-K linears with the
# round-3 epilogues, vs the default (round-2 interleaved-DMA loop, one tile per workgroup), same box, 30-iteration timings
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp UAV_EPI_ITERS_X=5
L=gpurun_out/r3_ab_short_k_linears_persistent.log
: > $L
for s in 0 1; do
  echo "UAV_CONV_PERSIST=$s" | tee -a $L
  UAV_CONV_PERSIST=$s timeout 60 python tools/bench_epilogue.py "linear 512->512 M=409600" "linear 2048->512 M=409600 bias+res32" "linear 512->1536" "linear 1024->1024" 2>&1 | grep '^{' | tee -a $L
done
