# round-end evidence in ONE gpurun call: rocprofv3 kernel stats of the default bench command, the PMC traffic pass,
# BASELINE configs[2] (propagation) and one configs[4] tile (348x384, video VAE) bench lines.  Only the summaries are
# kept under gpurun_out/ (the raw rocprofv3 databases exceed the 64-MiB merge limit).
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err
python $R/tools/rocpd_top_kernels.py $(find /tmp/prof_final -name "*.db" | head -1) $R/gpurun_out/rocprofv3_kernel_stats_bench.csv "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --no-cpu-baseline (MI355X, round 2; rocpd view top_kernels; 1 warmup + 2 timed clips)"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum \
  -d /tmp/pmc_traffic -o traffic -- python $R/bench.py --no-cpu-baseline --no-kernel-events --warmup 0 --steps 1 > $R/gpurun_out/pmc_traffic_bench.json 2> $R/gpurun_out/pmc_traffic.err
DB=$(find /tmp/pmc_traffic -name "*.db" | head -1); echo "db=$DB"
cd $R; python tools/pmc_traffic.py $DB > gpurun_out/pmc_conv_traffic_stdout.json 2>&1; cp profiles/pmc_conv_traffic.json gpurun_out/pmc_conv_traffic.json
timeout 300 python bench.py --propagation --no-cpu-baseline > gpurun_out/bench_config3_propagation.json 2> gpurun_out/bench_config3.err
timeout 400 python bench.py --video-vae --height 348 --width 384 --no-cpu-baseline > gpurun_out/bench_config5_tile.json 2> gpurun_out/bench_config5.err
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for f in bench_config3_propagation bench_config5_tile bench_default; do python -c "
import json; d=json.load(open('gpurun_out/$f.json')); print('$f', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1))"; done
du -sh gpurun_out
