#!/bin/bash
# round 4, GPU call 11: SQ counters of the rotated k-step (product, UAV_CONV_DMAV=6) vs the round-3 loop (=1) on the 3x3 layers
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
R=$PWD; export TMPDIR=/tmp
S=$R/gpurun_out/r4_run11_pmc_sq_rotated.jsonl; : > $S
cd /tmp
for c in c512_320 c256_320; do for v in 1 6; do
  rm -rf /tmp/pmc_sq
  UAV_CONV_DMAV=$v timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    -d /tmp/pmc_sq -o sq -- python $R/tools/bench_one.py $c 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_$c" "%conv_gemm256%" >> $S
  rm -rf /tmp/pmc_sq
  UAV_CONV_DMAV=$v timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/pmc_sq -o g -- python $R/tools/bench_one.py $c 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_${c}_pass2" "%conv_gemm256%" >> $S
done; done
cat $S
