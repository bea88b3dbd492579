#!/bin/bash
# round-end evidence, round 4, ONE gpurun call: GPU suite + smoke, rocprofv3 kernel stats of the default bench command, the PMC traffic
# pass (stamped with the kernel-source digest), SQ counters over the bench workload, the default line (CPU baseline + two-clip
# throughput mode), fp16-stream line, BASELINE configs[2], one configs[4] tile, 2 / 3 clips per GPU, the configs[3] schedule on one GPU
# with and without the two-stream overlap (peak memory of both on the line).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
R=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r4_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r4_tests_final.log
cp gpurun_out/parity.jsonl gpurun_out/r4_parity_final.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --no-cpu-baseline --no-throughput-mode > $R/gpurun_out/r4_bench_under_rocprof.json 2> $R/gpurun_out/r4_bench_under_rocprof.err
python $R/tools/rocpd_top_kernels.py $(find /tmp/prof_final -name "*.db" | head -1) $R/gpurun_out/r4_rocprofv3_kernel_stats_bench.csv "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --no-cpu-baseline --no-throughput-mode (MI355X, round 4 final; rocpd view top_kernels; 1 warmup + 2 timed clips + 1 instrumented clip)"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum \
  -d /tmp/pmc_traffic -o traffic -- python $R/bench.py --no-cpu-baseline --no-throughput-mode --no-kernel-events --warmup 0 --steps 1 > $R/gpurun_out/r4_pmc_traffic_bench.json 2> $R/gpurun_out/r4_pmc_traffic.err
DB=$(find /tmp/pmc_traffic -name "*.db" | head -1); echo "db=$DB"
cd $R; python tools/pmc_traffic.py $DB > gpurun_out/r4_pmc_conv_traffic_stdout.json 2>&1; cp profiles/pmc_conv_traffic.json gpurun_out/r4_pmc_conv_traffic.json
cd /tmp; rm -rf /tmp/pmc_sq
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
  -d /tmp/pmc_sq -o sq -- python $R/bench.py --no-cpu-baseline --no-throughput-mode --no-kernel-events --warmup 0 --steps 1 > /dev/null 2> $R/gpurun_out/r4_pmc_sq.err
for pat in "%conv_gemm256i_kernel<1, 0, 0>%" "%conv_gemm256i_kernel<1, 1%" "%conv_gemm256i_kernel<1, 2%" "%conv_gemm256i_kernel<1, 3%" "%conv_gemm256i%" "%attn512w%" "%gn_apply%"; do
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "bench_default" "$pat" | sed "s|^{|{\"kernels\": \"$pat\", |"
done > $R/gpurun_out/r4_pmc_sq_conv.jsonl 2>&1
cd $R
timeout 500 python bench.py > gpurun_out/r4_bench_default_final.json 2> gpurun_out/r4_bench_default_final.err
timeout 300 python bench.py --unet-stream f16 --no-cpu-baseline --no-throughput-mode > gpurun_out/r4_bench_f16_final.json 2> gpurun_out/r4_bench_f16_final.err
timeout 300 python bench.py --propagation --no-cpu-baseline --no-throughput-mode > gpurun_out/r4_bench_config3_final.json 2> gpurun_out/r4_bench_config3_final.err
timeout 400 python bench.py --video-vae --height 348 --width 384 --no-cpu-baseline --no-throughput-mode > gpurun_out/r4_bench_config5_tile_final.json 2> gpurun_out/r4_bench_config5_tile_final.err
timeout 300 python bench.py --clips-per-step 2 --no-cpu-baseline > gpurun_out/r4_bench_two_clips_final.json 2> gpurun_out/r4_bench_two_clips_final.err
timeout 300 python bench.py --clips-per-step 3 --no-cpu-baseline > gpurun_out/r4_bench_three_clips_final.json 2> gpurun_out/r4_bench_three_clips_final.err
timeout 400 python bench.py --shard-windows --frames 32 --warmup 0 --no-cpu-baseline > gpurun_out/r4_bench_config4_t32_1gpu_final.json 2> gpurun_out/r4_bench_config4_t32_1gpu_final.err
timeout 400 python bench.py --frames 32 --warmup 0 --no-cpu-baseline --no-throughput-mode --overlap-streams 2 > gpurun_out/r4_bench_t32_overlap2_final.json 2> gpurun_out/r4_bench_t32_overlap2_final.err
timeout 400 python bench.py --frames 32 --warmup 0 --no-cpu-baseline --no-throughput-mode --overlap-streams 0 > gpurun_out/r4_bench_t32_serial_final.json 2> gpurun_out/r4_bench_t32_serial_final.err
cat gpurun_out/r4_tests_final.log
for f in default_final f16_final config3_final config5_tile_final two_clips_final three_clips_final config4_t32_1gpu_final t32_overlap2_final t32_serial_final under_rocprof; do python -c "
import json; d=json.load(open('gpurun_out/r4_bench_$f.json')); r=d.get('roofline',{}); print('$f', round(d['value'],4), round(d['ms_per_step'],1), round(r.get('achieved',0),1), r.get('traffic'), d.get('cpu_baseline',{}).get('value'), d['config'].get('peak_memory_gb'), (d.get('throughput_mode') or {}).get('frames_per_s'))"; done
head -16 gpurun_out/r4_rocprofv3_kernel_stats_bench.csv | cut -c1-200
cat gpurun_out/r4_pmc_sq_conv.jsonl
du -sh gpurun_out
