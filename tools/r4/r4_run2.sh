#!/bin/bash
# round 4, GPU call 2: the EARLY-RELEASE k-loop (UAV_CONV_DMAV=4, product library) against the production loop (=1):
# digests, the conv kernel tests on the new loop, micro-benchmarks, clip A/B; SQ counters of the three loops (1, 4 and the
# rejected 8-phase candidate from the round-3 variant library) on the 3x3 512->512 @16x320x320 layer; then the headline-shape
# parity tests (configs[1] / configs[2] with consistent flows, sampler hi|lo operands on by default).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
R=$PWD; export TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=gpurun_out/r4_run2_ab_early_release.log; : > $L
export UAV_CONV_TILE=256
for v in 1 4 4; do echo "digests UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v timeout 120 python tools/conv_digest.py 2>&1 | tail -1 | tee -a $L; done
unset UAV_CONV_TILE
echo "conv kernel tests on UAV_CONV_DMAV=4" | tee -a $L
UAV_CONV_DMAV=4 timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "conv or gn or linear or upsample or shortcut" 2>&1 | tail -3 | tee -a $L
for v in 1 4; do echo "micro-benchmarks UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v UAV_EPI_ITERS_X=5 timeout 200 python tools/bench_epilogue.py 2>&1 | grep '^{' | tee -a $L; done
for v in 1 4 1 4; do
  UAV_CONV_DMAV=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --digest 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline',{})
print('UAV_CONV_DMAV=$v frames/s=%.4f ms/clip=%.1f conv TFLOP/s=%.1f sha=%s' % (d['value'], d['ms_per_step'], r.get('achieved',0), d['config']['output_sha256'][:16]))" | tee -a $L
done
# SQ counters, one pass per loop variant (own rocprofv3 runs, counters only with --kernel-trace)
S=$R/gpurun_out/r4_run2_pmc_sq_kloops.jsonl; : > $S
cd /tmp
for spec in "1:" "4:" "8:$R/tools/ab/libuav_hip_8phase.so"; do
  v=${spec%%:*}; lib=${spec#*:}
  rm -rf /tmp/pmc_sq
  UAV_HIP_LIB=$lib UAV_CONV_DMAV=$v timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    -d /tmp/pmc_sq -o sq -- python $R/tools/bench_one.py c512_320 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_c512_320" "%conv_gemm256%" >> $S
  rm -rf /tmp/pmc_sq
  UAV_HIP_LIB=$lib UAV_CONV_DMAV=$v timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/pmc_sq -o g -- python $R/tools/bench_one.py c512_320 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_c512_320_pass2" "%conv_gemm256%" >> $S
done
cd $R
cat $S
timeout 900 python -m pytest tests/test_parity_r4_gpu.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r4_run2_parity_tests.log
cp gpurun_out/parity.jsonl gpurun_out/r4_run2_parity.jsonl
cat gpurun_out/r4_run2_parity_tests.log
python - <<'PY'
import json
for l in open('gpurun_out/r4_run2_parity.jsonl'):
    d = json.loads(l)
    c = d.pop('latents_rel_l2_per_step', None)
    if c: d['curve_1_5_10_20_24_26_28_30'] = [c[0], c[4], c[9], c[19], c[23], c[25], c[27], c[29]]
    print(d)
PY
