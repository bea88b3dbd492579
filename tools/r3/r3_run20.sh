#!/bin/bash
# round 3, GPU call 20: two ranks on ONE GPU (gloo transport) — sharded (window x guidance branch) schedule vs the 1-rank run, same bits
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multigpu_gpu.py -q -m gpu -rs 2>&1 | tail -12 | tee gpurun_out/r3_multigpu_two_ranks_one_gpu.log
