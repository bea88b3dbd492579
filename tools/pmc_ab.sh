mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for ko in 0 1; do
  export UAV_CONV_KORDER=$ko
  echo "korder=$ko"; timeout 100 python $R/tools/bench_kernels.py conv 2>&1 | grep -E "3x3 512->512 @16x320|3x3 256->256|t5 512|3x3 1024->512" 
  timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/gpurun_out/pmc_l2_ko$ko -o l2 -- python $R/tools/bench_one.py c512_320 3 > /dev/null 2>&1
done
