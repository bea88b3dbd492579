# HBM-side traffic of the conv kernels over the default bench workload (separate PMC pass, kernel-trace only).
# Writes gpurun_out/pmc_traffic/*.db; tools/pmc_traffic.py turns it into profiles/pmc_conv_traffic.json.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum \
  -d $R/gpurun_out/pmc_traffic -o traffic -- python $R/bench.py --no-cpu-baseline --no-kernel-events --warmup 0 --steps 1 > $R/gpurun_out/pmc_traffic_bench.json 2> $R/gpurun_out/pmc_traffic.err
echo "rc=$?"; ls -la $R/gpurun_out/pmc_traffic 2>/dev/null | tail -3
