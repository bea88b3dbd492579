# HBM-side traffic of the conv kernels over the default bench workload (separate PMC passes, kernel-trace only).
# Writes gpurun_out/pmc_traffic*/ (rocpd databases); `python tools/pmc_traffic.py <db> [<db> ...]` turns them into
# profiles/pmc_conv_traffic.json.  One four-counter pass; if rocprofv3 dies in it (seen once in round 5: SIGSEGV inside the
# first launch), two two-counter passes are taken instead.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-kernel-events --no-throughput-mode --warmup 0 --steps 1"
pass() {   # pass <dir suffix> <counters...>
  local sfx=$1; shift
  rm -rf $R/gpurun_out/pmc_traffic$sfx
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_traffic$sfx -o traffic -- $BENCH \
    > $R/gpurun_out/pmc_traffic${sfx}_bench.json 2> $R/gpurun_out/pmc_traffic$sfx.err
  local rc=$?
  echo "pass '$sfx' ($*): rc=$rc"; ls -la $R/gpurun_out/pmc_traffic$sfx 2>/dev/null | tail -2
  return $rc
}
if ! pass "" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum; then
  rm -rf $R/gpurun_out/pmc_traffic
  pass _rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pass _wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
fi
find $R/gpurun_out -name "*.db" -path "*pmc_traffic*" | xargs -r python $R/tools/pmc_traffic.py --out $R/gpurun_out/pmc_conv_traffic.json 2>&1 | tail -25
# the reduced json is what travels back (gpurun_out/ is capped at 64 MiB): drop the databases once it exists
[ -s $R/gpurun_out/pmc_conv_traffic.json ] && rm -rf $R/gpurun_out/pmc_traffic $R/gpurun_out/pmc_traffic_rd $R/gpurun_out/pmc_traffic_wr
