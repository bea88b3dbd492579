# round-end evidence in ONE gpurun call: rocprofv3 kernel stats of the default bench command, the PMC traffic pass,
# BASELINE configs[2] (propagation) and one configs[4] tile (348x384, video VAE) bench lines
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err
find $R/gpurun_out/prof_final -name "*kernel_stats*" | head -3
bash $R/tools/pmc_traffic.sh
cd $R
timeout 300 python bench.py --propagation --no-cpu-baseline > gpurun_out/bench_config3_propagation.json 2> gpurun_out/bench_config3.err
timeout 400 python bench.py --video-vae --height 348 --width 384 --no-cpu-baseline > gpurun_out/bench_config5_tile.json 2> gpurun_out/bench_config5.err
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for f in bench_config3_propagation bench_config5_tile bench_default; do python -c "
import json; d=json.load(open('gpurun_out/$f.json')); print('$f', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1))"; done
