#!/bin/bash
# One parameterised script for every GPU call of a round (VERDICT r4 hygiene #9: replaces tools/r4/r4_run*.sh).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/run.sh TAG step [step ...]'
# TAG names the output files under gpurun_out/ (copy what is kept to profiles/rNN_*).  Steps (run in the order given):
#   ktests          conv / linear kernel tests (tests/test_kernels_gpu.py -k "conv or linear or shortk or w4 or phase or shortcut or broadcast or groupnorm_stat")
#   tests           the whole GPU suite + __graft_entry__.smoke()
#   shortk          tools/bench_shortk.py, asm-scheduled k-step;   shortk2: the compiler-scheduled one (UAV_CONV_SK=2)
#   epi             tools/bench_epilogue.py
#   calib           tools/calib_gemm.py (plain GEMM vs conv 1x1 vs conv 3x3);   calib_pmc: its SQ / GRBM counter passes
#   bench           python bench.py (default line);   bench_sk0: the same with UAV_CONV_SK=0 (same-box A/B)
#   bench_driver    python bench.py --steps 20 --warmup 2 (the way the driver runs it)
#   bench_prop / bench_high / bench_f16   one clip of --propagation (configs[2]) / --precision high / --unet-stream f16
#   tests_all       the GPU suite without -x (every failure in one call)
#   bench1          python bench.py --steps 1 --no-cpu-baseline (quick line with the kernel breakdown)
#   prof            rocprofv3 --kernel-trace --stats over bench.py --steps 2
#   traffic         tools/pmc_traffic.sh (conv HBM bytes per launch, own --pmc passes)
#   digest          tools/conv_digest.py
#   bench1_head     bench1 on tools/ab/libuav_hip_head.so (a build of the library from HEAD's sources, same-box A/B of uncommitted kernel work)
#   trace0 / trace_small   phase stamps of the four-wave kernel on the K = 512 linears (full grid / 64..3200-tile grids)
#   xattn / xtests  fused cross-attention sub-layer: tools/bench_xattn.py (vs the four-launch chain) / its kernel tests;  bench1_nox: bench1 with UAV_XATTN_FUSED=0
#   t32             configs[3] at its real length: T = 32, 320x320, 30 steps vs the GPU oracle (tests/test_parity_r6_gpu.py, ~12 min)
#   bench1_cfgsplit one 8-frame clip with its two guidance branches on two HIP streams
#   bench_noev      event-overhead A/B: default vs --no-kernel-events, twice interleaved
#   bench_c3 / bench_c4 / bench_2clips   configs[3] schedule on one GPU / one configs[4] tile with vae_video / two clips per GPU
#   xpmc / attn512   SQ counters of the fused transformer / attention kernels (own --pmc pass);  d = 512 attention at L = 102 400, ring kernel vs attn512w
#   bench1_lnfold / parity_lnfold   the LayerNorm-fold switch (UAV_LN_FOLD=1): clip time and the headline parity test
# Environment: any UAV_* variable is passed through to every step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# steps that reach development kernels (short-K, stamped four-wave conv instance, LayerNorm fold, ablation builds) load the development
# build of the library (tools/ab/build_dev.sh all) instead of the product one
DEV="UAV_HIP_LIB=$R/tools/ab/libuav_hip_dev.so"
log() { echo "== $(date +%H:%M:%S) $*" | tee -a $O/${TAG}_steps.log; }
for step in "$@"; do
  log "$step"
  case $step in
    ktests)   timeout 600 python -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -n 2 -k "conv or linear or shortk or w4 or phase or shortcut or broadcast or groupnorm_stat" 2>&1 | tail -5 | tee $O/${TAG}_ktests.log ;;
    tests)    rm -f $O/parity.jsonl; timeout 1800 python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -15 | tee $O/${TAG}_tests.log; cp $O/parity.jsonl $O/${TAG}_parity_full_suite.jsonl 2> /dev/null
              (cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/${TAG}_tests.log) ;;
    tests_all) rm -f $O/parity.jsonl; timeout 1800 python -m pytest $R/tests -m gpu -q 2>&1 | tail -40 | tee $O/${TAG}_tests.log; cp $O/parity.jsonl $O/${TAG}_parity_full_suite.jsonl 2> /dev/null
              (cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/${TAG}_tests.log) ;;
    shortk) env $DEV   timeout 400 python $R/tools/bench_shortk.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_shortk.log ;;
    shortk2) env $DEV  UAV_CONV_SK=2 timeout 400 python $R/tools/bench_shortk.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_shortk_compiler_kstep.log ;;
    trace) env $DEV    UAV_CONV_W4_TRACE=1 timeout 300 python $R/tools/trace_w4.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_w4_trace.log ;;
    trace_small) env $DEV UAV_TRACE_SMALL=1 UAV_CONV_TILE=256 UAV_CONV_W4_MINK=0 UAV_CONV_W4_TRACE=1 timeout 300 python $R/tools/trace_w4.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_w4_trace_small_grids.log ;;
    bench1_lnfold) (cd $R && env $DEV UAV_LN_FOLD=1 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_lnfold.json) ;;
    parity_lnfold) (cd $R && rm -f gpurun_out/parity.jsonl; env $DEV UAV_LN_FOLD=1 timeout 900 python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -k "headline" 2>&1 | tail -15 | tee $O/${TAG}_parity_lnfold.log; mv gpurun_out/parity.jsonl $O/${TAG}_parity_lnfold.jsonl 2> /dev/null) ;;
    bench1_nt) (cd $R && UAV_HIP_LIB=$R/tools/ab/libuav_hip_nt.so timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee -a $O/${TAG}_bench1_nt_lib.json) ;;
    bench1_rep) (cd $R && timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee -a $O/${TAG}_bench1_rep.json) ;;
    mink_sweep) for mk in ${UAV_MINKS:-0 256 512 1024}; do (cd $R && UAV_CONV_W4_MINK=$mk timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); kb = d['kernel_breakdown']
print(json.dumps({'UAV_CONV_W4_MINK': $mk, 'frames_per_s': round(d['value'], 4), 'ms_per_clip': round(d['ms_per_step'], 1), 'conv_ms': kb['conv_gemm']['ms'], 'conv_tflops': kb['conv_gemm']['tflops']}))" | tee -a $O/${TAG}_w4_mink_sweep.jsonl); done ;;
    bench1_f32lds) (cd $R && UAV_HIP_LIB=$R/tools/ab/libuav_hip_f32lds.so timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee -a $O/${TAG}_bench1_f32lds_lib.json) ;;
    atests)   timeout 600 python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_clip_text_gpu.py -m gpu -x -q -k "attention or clip" 2>&1 | tail -5 | tee $O/${TAG}_attention_tests.log ;;
    bench1_prev) (cd $R && UAV_HIP_LIB=$R/tools/ab/libuav_hip_prev.so timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee -a $O/${TAG}_bench1_prev_lib.json) ;;
    bench1_head) (cd $R && UAV_HIP_LIB=$R/tools/ab/libuav_hip_head.so timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_head_lib.json) ;;
    trace0) env $DEV   UAV_CONV_W4_MINK=0 UAV_CONV_W4_TRACE=1 timeout 300 python $R/tools/trace_w4.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_w4_trace.log ;;
    w4ab)     timeout 500 python $R/tools/bench_w4.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_w4_vs_8wave.log ;;
    epi)      timeout 400 python $R/tools/bench_epilogue.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_epilogue.log ;;
    epi_w40)  UAV_CONV_W4=0 timeout 400 python $R/tools/bench_epilogue.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_epilogue_8wave.log ;;
    calib_w40) UAV_CONV_W4=0 timeout 300 python $R/tools/calib_gemm.py conv1x1 conv3x3 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_calib_8wave.jsonl ;;
    bench1_mink) (cd $R && UAV_CONV_W4_MINK=${UAV_MINK:-1024} timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_w4_mink.json) ;;
    bench1_w40) (cd $R && UAV_CONV_W4=0 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_8wave.json) ;;
    calib)    timeout 300 python $R/tools/calib_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_calib.jsonl ;;
    blas)     timeout 300 python $R/tools/calib_blas_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_calib_blas_shapes.jsonl ;;
    w4_pmc)     # SQ / clock counters of the product conv kernel (four-wave) on the two calibration arms; one SQ pass + one clock pass per arm
      L=$O/${TAG}_w4_pmc.jsonl; : > $L
      for arm in conv3x3 conv1x1; do
        rm -rf /tmp/pmc_c
        UAV_CALIB_ITERS=2 timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
          -d /tmp/pmc_c -o sq -- python $R/tools/calib_gemm.py $arm > /dev/null 2>&1
        python $R/tools/pmc_reduce.py "$(find /tmp/pmc_c -name '*.db' | head -1)" "${arm}_w4_sq" "%conv_gemm256w%" >> $L
        rm -rf /tmp/pmc_c
        UAV_CALIB_ITERS=2 timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d /tmp/pmc_c -o g -- python $R/tools/calib_gemm.py $arm > /dev/null 2>&1
        python $R/tools/pmc_reduce.py "$(find /tmp/pmc_c -name '*.db' | head -1)" "${arm}_w4_clock" "%conv_gemm256w%" >> $L
      done
      cat $L ;;
    calib_pmc)
      L=$O/${TAG}_calib_pmc.jsonl; : > $L
      for arm in blas conv1x1 conv3x3; do
        case $arm in blas) pat="%Cijk%";; *) pat="%conv_gemm256i%";; esac
        rm -rf /tmp/pmc_c
        UAV_CALIB_ITERS=2 timeout 240 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
          -d /tmp/pmc_c -o sq -- python $R/tools/calib_gemm.py $arm > /dev/null 2>&1
        db="$(find /tmp/pmc_c -name '*.db' | head -1)"
        python $R/tools/pmc_reduce.py "$db" "${arm}_sq" "$pat" >> $L
        python - "$db" >> $L <<'PY'
import json, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
try:
    rows = list(cur.execute("select name, count(*), avg(duration) from kernels group by name order by sum(duration) desc limit 4"))
    print(json.dumps({"top_kernels": [[r[0][:90], r[1], r[2]] for r in rows]}))
except Exception as e:
    print(json.dumps({"top_kernels_error": str(e)}))
PY
        rm -rf /tmp/pmc_c
        UAV_CALIB_ITERS=2 timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d /tmp/pmc_c -o g -- python $R/tools/calib_gemm.py $arm > /dev/null 2>&1
        python $R/tools/pmc_reduce.py "$(find /tmp/pmc_c -name '*.db' | head -1)" "${arm}_clock" "$pat" >> $L
      done
      cat $L ;;
    xattn)    timeout 300 python $R/tools/bench_xattn.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_xattn_fused_vs_chain.jsonl ;;
    xtrace)   timeout 300 python $R/tools/trace_xattn.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_xattn_phase_trace.jsonl ;;
    xtests)   timeout 600 python -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -k "fused_cross or fused_temporal or fused_block or attention or hilo or fused_sublayer or fused_feed or whole_transformer or whole_block" 2>&1 | tail -8 | tee $O/${TAG}_xattn_tests.log ;;
    bench1_tail) (cd $R && UAV_TAIL_HILO=0 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_tail_hilo_off.json) ;;
    parity_head) (cd $R && rm -f gpurun_out/parity.jsonl; timeout 900 python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -k "headline" 2>&1 | tail -6 | tee $O/${TAG}_parity_headline.log; mv gpurun_out/parity.jsonl $O/${TAG}_parity_headline.jsonl 2> /dev/null) ;;
    parity_head_tail) (cd $R && rm -f gpurun_out/parity.jsonl; UAV_TAIL_HILO=0 timeout 900 python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -k "headline" 2>&1 | tail -6 | tee $O/${TAG}_parity_headline_tail_hilo_off.log; mv gpurun_out/parity.jsonl $O/${TAG}_parity_headline_tail_hilo_off.jsonl 2> /dev/null) ;;
    bench1_nox) (cd $R && UAV_XATTN_FUSED=0 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_four_launch_chain.json) ;;
    t32)      (cd $R && rm -f gpurun_out/parity.jsonl; UAV_PARITY_T32=1 timeout 1500 python -m pytest tests/test_parity_r6_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $O/${TAG}_parity_t32.log; mv gpurun_out/parity.jsonl $O/${TAG}_parity_configs3_t32_320.jsonl 2> /dev/null) ;;
    bench1_cfgsplit) (cd $R && timeout 400 python bench.py --overlap-streams 2 --overlap-split-cfg --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench1_cfg_branches_on_two_streams.json) ;;
    bench_noev) (cd $R && for i in 1 2; do timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee -a $O/${TAG}_event_overhead_ab_default.jsonl; timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-kernel-events 2> /dev/null | tee -a $O/${TAG}_event_overhead_ab_no_kernel_events.jsonl; done) ;;
    bench_c3)  (cd $R && timeout 900 python bench.py --frames 32 --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench_configs3_schedule_t32_one_gpu.json) ;;
    bench_c4)  (cd $R && timeout 900 python bench.py --video-vae --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench_configs4_tile_video_vae.json) ;;
    bench_2clips) (cd $R && timeout 900 python bench.py --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline 2> /dev/null | tee $O/${TAG}_bench_two_clips_per_gpu.json) ;;
    bench)    (cd $R && timeout 900 python bench.py 2> $O/${TAG}_bench.err | tee $O/${TAG}_bench.json) ;;
    bench_driver) (cd $R && timeout 1200 python bench.py --steps 20 --warmup 2 2> /dev/null | tee $O/${TAG}_bench_driver_style_steps20.json) ;;
    bench_prop) (cd $R && timeout 600 python bench.py --propagation --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench_configs2_propagation.json) ;;
    bench_high) (cd $R && timeout 600 python bench.py --precision high --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench_precision_high.json) ;;
    bench_f16)  (cd $R && timeout 600 python bench.py --unet-stream f16 --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> /dev/null | tee $O/${TAG}_bench_unet_stream_f16.json) ;;
    bench_sk0) (cd $R && UAV_CONV_SK=0 timeout 900 python bench.py --no-cpu-baseline 2> /dev/null | tee $O/${TAG}_bench_sk0.json) ;;
    bench1)   (cd $R && timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> $O/${TAG}_bench1.err | tee $O/${TAG}_bench1.json) ;;
    bench1_sk0) (cd $R && UAV_CONV_SK=0 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> /dev/null | tee $O/${TAG}_bench1_sk0.json) ;;
    shape)    (cd $R && UAV_BENCH_DETAIL=1 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode 2> $O/${TAG}_per_shape.txt > $O/${TAG}_bench_detail.json; grep -v amdgpu.ids $O/${TAG}_per_shape.txt | head -70) ;;
    prof)     rm -rf /tmp/prof; (cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode > $O/${TAG}_bench_under_rocprof.json 2> /dev/null)
              # ROCm 7.2 rocprofv3 writes a rocpd database, not a csv: its `top_kernels` view IS the --stats kernel summary
              db=$(find /tmp/prof -name "*.db" | head -1)
              [ -n "$db" ] && python $R/tools/rocpd_top_kernels.py "$db" $O/${TAG}_rocprofv3_kernel_stats.csv "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode (MI355X; rocpd view top_kernels; 1 warm-up + 2 timed clips + 1 instrumented clip, all serial)" && head -14 $O/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-200 ;;
    traffic)  (cd $R && bash tools/pmc_traffic.sh 2>&1 | tail -5); cp $O/pmc_conv_traffic.json $O/${TAG}_pmc_conv_traffic.json 2> /dev/null ;;
    xpmc)     # SQ counters of the round-6 kernels: matrix-pipe busy, parked / stalled / issuing wave cycles, LDS activity and bank conflicts
      L=$O/${TAG}_pmc_sq_fused_kernels.jsonl; : > $L
      C="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
      rm -rf /tmp/pmc_x; UAV_XA_ROUNDS=1 UAV_XA_PER=2 timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_x -o sq -- python $R/tools/bench_xattn.py > /dev/null 2>&1
      db="$(find /tmp/pmc_x -name '*.db' | head -1)"
      for pat in "tattn_sublayer_kernel<2, 1, 0>" "tattn_sublayer_kernel<2, 0, 0>" "tattn_sublayer_kernel<0, 0, 0>" "ff_sublayer_kernel" "xattn_sublayer_kernel<0>" "conv_gemm256w_kernel"; do
        python $R/tools/pmc_reduce.py "$db" "$pat" "%$pat%" >> $L; done
      rm -rf /tmp/pmc_x; UAV_XA_ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_x -o sq -- python $R/tools/bench_attn512.py > /dev/null 2>&1
      python $R/tools/pmc_reduce.py "$(find /tmp/pmc_x -name '*.db' | head -1)" "attn512x_kernel" "%attn512x_kernel%" >> $L
      rm -rf /tmp/pmc_x; UAV_ATTN512X=0 UAV_XA_ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_x -o sq -- python $R/tools/bench_attn512.py > /dev/null 2>&1
      python $R/tools/pmc_reduce.py "$(find /tmp/pmc_x -name '*.db' | head -1)" "attn512w_kernel" "%attn512w_kernel%" >> $L
      cat $L ;;
    attn512)  timeout 300 python $R/tools/bench_attn512.py 3 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_attn512.jsonl; UAV_ATTN512X=0 timeout 300 python $R/tools/bench_attn512.py 3 2>&1 | grep -v amdgpu.ids | tee -a $O/${TAG}_attn512.jsonl ;;
    digest)   timeout 200 python $R/tools/conv_digest.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_digest.log ;;
    *)        echo "unknown step $step" ;;
  esac
done
log done
