#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/host_issue_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_host_issue_time.log
