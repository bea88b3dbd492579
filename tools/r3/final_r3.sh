#!/bin/bash
# round-end evidence, round 3, ONE gpurun call: GPU suite + smoke, rocprofv3 kernel stats of the default bench command, the PMC
# traffic pass (stamped with the kernel-source digest), default line (with the CPU baseline), fp16-stream line, BASELINE
# configs[2] (propagation), one configs[4] tile, the 2-clips serving mode.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
R=$PWD; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r3_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r3_tests_final.log
cp gpurun_out/parity.jsonl gpurun_out/r3_parity_final.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/gpurun_out/r3_bench_under_rocprof.json 2> $R/gpurun_out/r3_bench_under_rocprof.err
python $R/tools/rocpd_top_kernels.py $(find /tmp/prof_final -name "*.db" | head -1) $R/gpurun_out/r3_rocprofv3_kernel_stats_bench.csv "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --no-cpu-baseline (MI355X, round 3 final; rocpd view top_kernels; 1 warmup + 2 timed clips + 1 instrumented clip)"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum \
  -d /tmp/pmc_traffic -o traffic -- python $R/bench.py --no-cpu-baseline --no-kernel-events --warmup 0 --steps 1 > $R/gpurun_out/r3_pmc_traffic_bench.json 2> $R/gpurun_out/r3_pmc_traffic.err
DB=$(find /tmp/pmc_traffic -name "*.db" | head -1); echo "db=$DB"
cd $R; python tools/pmc_traffic.py $DB > gpurun_out/r3_pmc_conv_traffic_stdout.json 2>&1; cp profiles/pmc_conv_traffic.json gpurun_out/r3_pmc_conv_traffic.json
# SQ counters of the conv kernels over the same workload (own pass): MFMA pipe busy, parked wave cycles, LDS activity / conflicts
cd /tmp; rm -rf /tmp/pmc_sq
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
  -d /tmp/pmc_sq -o sq -- python $R/bench.py --no-cpu-baseline --no-kernel-events --warmup 0 --steps 1 > /dev/null 2> $R/gpurun_out/r3_pmc_sq.err
for pat in "%conv_gemm256i_kernel<1, 0, 0>%" "%conv_gemm256i_kernel<1, 1%" "%conv_gemm256i_kernel<1, 2%" "%conv_gemm256i_kernel<1, 3%" "%conv_gemm256i%" "%attn512w%" "%gn_apply%"; do
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "bench_default" "$pat" | sed "s|^{|{\"kernels\": \"$pat\", |"
done > $R/gpurun_out/r3_pmc_sq_conv.jsonl 2>&1
cd $R
timeout 400 python bench.py > gpurun_out/r3_bench_default_final.json 2> gpurun_out/r3_bench_default_final.err
timeout 300 python bench.py --unet-stream f16 --no-cpu-baseline > gpurun_out/r3_bench_f16_final.json 2> gpurun_out/r3_bench_f16_final.err
timeout 300 python bench.py --propagation --no-cpu-baseline > gpurun_out/r3_bench_config3_final.json 2> gpurun_out/r3_bench_config3_final.err
timeout 400 python bench.py --video-vae --height 348 --width 384 --no-cpu-baseline > gpurun_out/r3_bench_config5_tile_final.json 2> gpurun_out/r3_bench_config5_tile_final.err
timeout 300 python bench.py --clips-per-step 2 --no-cpu-baseline > gpurun_out/r3_bench_two_clips_final.json 2> gpurun_out/r3_bench_two_clips_final.err
timeout 400 python bench.py --shard-windows --frames 32 --warmup 0 --no-cpu-baseline > gpurun_out/r3_bench_config4_t32_1gpu_final.json 2> gpurun_out/r3_bench_config4_t32_1gpu_final.err
cat gpurun_out/r3_tests_final.log
for f in default_final f16_final config3_final config5_tile_final two_clips_final config4_t32_1gpu_final under_rocprof; do python -c "
import json; d=json.load(open('gpurun_out/r3_bench_$f.json')); r=d.get('roofline',{}); print('$f', round(d['value'],4), round(d['ms_per_step'],1), round(r.get('achieved',0),1), r.get('traffic'), d.get('cpu_baseline',{}).get('value'))"; done
head -14 gpurun_out/r3_rocprofv3_kernel_stats_bench.csv
cat gpurun_out/r3_pmc_sq_conv.jsonl
du -sh gpurun_out
