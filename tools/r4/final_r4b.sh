#!/bin/bash
# round 4, last evidence call (the rotated k-step became the default after final_r4.sh had run): GPU suite + smoke of the final tree,
# rocprofv3 kernel stats, PMC traffic pass (the conv sources changed: new digest), default line, fp16-stream line, configs[2].
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
R=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12 > gpurun_out/r4b_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r4b_tests_final.log
cp gpurun_out/parity.jsonl gpurun_out/r4b_parity_final.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --no-cpu-baseline --no-throughput-mode > $R/gpurun_out/r4b_bench_under_rocprof.json 2> $R/gpurun_out/r4b_bench_under_rocprof.err
python $R/tools/rocpd_top_kernels.py $(find /tmp/prof_final -name "*.db" | head -1) $R/gpurun_out/r4b_rocprofv3_kernel_stats_bench.csv "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --no-cpu-baseline --no-throughput-mode (MI355X, round 4 final tree: rotated k-step default; rocpd view top_kernels; 1 warmup + 2 timed clips + 1 instrumented clip)"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum \
  -d /tmp/pmc_traffic -o traffic -- python $R/bench.py --no-cpu-baseline --no-throughput-mode --no-kernel-events --warmup 0 --steps 1 > $R/gpurun_out/r4b_pmc_traffic_bench.json 2> $R/gpurun_out/r4b_pmc_traffic.err
DB=$(find /tmp/pmc_traffic -name "*.db" | head -1)
cd $R; python tools/pmc_traffic.py $DB > gpurun_out/r4b_pmc_conv_traffic_stdout.json 2>&1; cp profiles/pmc_conv_traffic.json gpurun_out/r4b_pmc_conv_traffic.json
timeout 500 python bench.py > gpurun_out/r4b_bench_default_final.json 2> gpurun_out/r4b_bench_default_final.err
timeout 300 python bench.py --unet-stream f16 --no-cpu-baseline --no-throughput-mode > gpurun_out/r4b_bench_f16_final.json 2> /dev/null
timeout 300 python bench.py --propagation --no-cpu-baseline --no-throughput-mode > gpurun_out/r4b_bench_config3_final.json 2> /dev/null
UAV_CONV_DMAV=1 timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r4b_bench_default_dmav1.json 2> /dev/null
cat gpurun_out/r4b_tests_final.log
for f in default_final f16_final config3_final default_dmav1 under_rocprof; do python -c "
import json; d=json.load(open('gpurun_out/r4b_bench_$f.json')); r=d.get('roofline',{}); print('$f', round(d['value'],4), round(d['ms_per_step'],1), round(r.get('achieved',0),1), r.get('frac'), r.get('traffic'), d.get('cpu_baseline',{}).get('value'), (d.get('throughput_mode') or {}).get('frames_per_s'))"; done
head -12 gpurun_out/r4b_rocprofv3_kernel_stats_bench.csv | cut -c1-180
