# round-end evidence, round 2 final state, ONE gpurun call: GPU suite + smoke, rocprofv3 kernel stats of the default bench
# command, the PMC traffic pass, BASELINE configs[2] / one configs[4] tile, the default line and the 2-clips serving mode.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r2_tests33.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r2_tests33.log
cp gpurun_out/parity.jsonl gpurun_out/r2_parity33.jsonl
bash tools/final_profiles.sh > gpurun_out/final_profiles.log 2>&1
timeout 300 python bench.py --clips-per-step 2 --no-cpu-baseline > gpurun_out/bench_two_clips.json 2> gpurun_out/bench_two_clips.err
cat gpurun_out/r2_tests33.log; tail -5 gpurun_out/final_profiles.log
python -c "
import json; d=json.load(open('gpurun_out/bench_two_clips.json')); print('two clips', round(d['value'],4), round(d['ms_per_step'],1))"
head -12 gpurun_out/rocprofv3_kernel_stats_bench.csv
