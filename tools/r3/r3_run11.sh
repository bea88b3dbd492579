#!/bin/bash
# round 3, GPU call 11: what keeping the conv1 -> norm2 branch tensor in fp16 (UAV_BRANCH_F32=0) does to the parity numbers
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
UAV_BRANCH_F32=0 timeout 900 python -m pytest tests/test_parity_r3_gpu.py tests/test_video_io_gpu.py -q -m gpu 2>&1 | tail -12
grep -E "r3_" gpurun_out/parity.jsonl | cut -c1-700
