#!/bin/bash
# round 3, GPU call 28: fabric-side bytes per launch of the K = 512 linears with the fp32-stream epilogues (PMC pass over the
# micro-benchmark; algorithmic: plain 0.84 GB, +res16 1.26, +res32->f32 2.10, +res32->f16 1.68)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 100 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum \
  -d /tmp/pmc_case -o case -- python $R/tools/bench_epilogue.py "linear 512->512 M=409600 bias" > $R/gpurun_out/r3_pmc_case_stdout.txt 2> $R/gpurun_out/r3_pmc_case.err
python $R/tools/pmc_case.py $(find /tmp/pmc_case -name "*.db" | head -1) 8 | tee $R/gpurun_out/r3_pmc_linear_epilogue_traffic.jsonl
grep '^{' $R/gpurun_out/r3_pmc_case_stdout.txt | tee -a $R/gpurun_out/r3_pmc_linear_epilogue_traffic.jsonl
tail -3 $R/gpurun_out/r3_pmc_case.err
