#!/bin/bash
# round 4, GPU call 12: the FOUR-STAGE RING candidate (tools/r4/conv_gemm_ring4.patch, UAV_CONV_DMAV=7 in the variant library
# tools/ab/libuav_hip_ring.so) against the product loop (=6): digests, conv kernel tests on it, micro-benchmarks, clip A/B, SQ counters.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
R=$PWD; export TMPDIR=/tmp UAV_HIP_LIB=$R/tools/ab/libuav_hip_ring.so
L=gpurun_out/r4_run12_ab_ring4.log; : > $L
export UAV_CONV_TILE=256
for v in 6 7 7; do echo "digests UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v timeout 100 python tools/conv_digest.py 2>&1 | tail -1 | tee -a $L; done
unset UAV_CONV_TILE
echo "conv kernel tests on UAV_CONV_DMAV=7" | tee -a $L
UAV_CONV_DMAV=7 timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "conv or gn or linear or upsampl or shortcut" 2>&1 | tail -3 | tee -a $L
for v in 6 7; do echo "micro-benchmarks UAV_CONV_DMAV=$v" | tee -a $L; UAV_CONV_DMAV=$v UAV_EPI_ITERS_X=4 timeout 150 python tools/bench_epilogue.py 2>&1 | grep '^{' | tee -a $L; done
for v in 6 7 6 7; do
  UAV_CONV_DMAV=$v timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --digest 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline',{})
print('UAV_CONV_DMAV=$v frames/s=%.4f ms/clip=%.1f conv TFLOP/s=%.1f sha=%s' % (d['value'], d['ms_per_step'], r.get('achieved',0), d['config']['output_sha256'][:16]))" | tee -a $L
done
S=$R/gpurun_out/r4_run12_pmc_sq_ring4.jsonl; : > $S
cd /tmp
for v in 6 7; do
  rm -rf /tmp/pmc_sq
  UAV_CONV_DMAV=$v timeout 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    -d /tmp/pmc_sq -o sq -- python $R/tools/bench_one.py c512_320 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_c512_320" "%conv_gemm256%" >> $S
  rm -rf /tmp/pmc_sq
  UAV_CONV_DMAV=$v timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/pmc_sq -o g -- python $R/tools/bench_one.py c512_320 3 > /dev/null 2>&1
  python $R/tools/pmc_reduce.py $(find /tmp/pmc_sq -name "*.db" | head -1) "dmav${v}_c512_320_pass2" "%conv_gemm256%" >> $S
done
cat $S
