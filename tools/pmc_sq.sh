# SQ counters of the 256x256 conv kernel on the 3x3 512->512 @16x320x320 layer and the K=512 linear, round-1 loop
# (UAV_CONV_DMAV=0) vs interleaved DMA (=1): MFMA pipe busy, parked / issue-stalled / issuing wave cycles, LDS activity.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; L=$R/gpurun_out/pmc_sq.jsonl; : > $L
for v in 0 1; do for c in c512_320 lin512; do
  rm -rf /tmp/pmc_sq
  UAV_CONV_DMAV=$v timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    -d /tmp/pmc_sq -o sq -- python $R/tools/bench_one.py $c 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_$c" "%conv_gemm256%" >> $L
done; done
for v in 0 1; do
  rm -rf /tmp/pmc_sq
  UAV_CONV_DMAV=$v timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d /tmp/pmc_sq -o g -- python $R/tools/bench_one.py c512_320 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_c512_320_clock" "%conv_gemm256%" >> $L
done
cat $L
