#!/bin/bash
# round 4, GPU call 3: (a) up-sampler statistics in a shared workspace: kernel tests + clip A/B (UAV_FUSE_UPSAMPLE_GN=0|1);
# (b) what the hi|lo operand knobs COST per clip (UAV_SAMPLER_HILO / UAV_TAIL_HILO, same box); (c) configs[2] against the
# reference pipeline's own fixture; (d) the precision table at the headline shape with the tail knob; (e) default bench line
# with the two-clip throughput leg.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
R=$PWD; export TMPDIR=/tmp
L=gpurun_out/r4_run3.log; : > $L
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "upsampl or gn or stat" 2>&1 | tail -3 | tee -a $L
one() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --digest 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline',{}); kb=d.get('kernel_breakdown',{})
print('$label frames/s=%.4f ms/clip=%.1f conv TFLOP/s=%.1f conv_ms=%.0f gn_stats_ms=%s gn_apply_ms=%s cast_ms=%s sha=%s' % (d['value'], d['ms_per_step'], r.get('achieved',0), kb.get('conv_gemm',{}).get('ms',0), kb.get('groupnorm_stats',{}).get('ms'), kb.get('groupnorm_apply',{}).get('ms'), kb.get('cast_f16',{}).get('ms'), d['config']['output_sha256'][:16]))" | tee -a $L
}
for rep in 1 2; do
  one "default(sampler_hilo=1,up_gn=1)" UAV_X=0
  one "up_gn=0" UAV_FUSE_UPSAMPLE_GN=0
  one "sampler_hilo=0" UAV_SAMPLER_HILO=0
  one "sampler_hilo=1+tail_hilo=1" UAV_TAIL_HILO=1
done
UAV_R4_SAVE_ORACLE=/tmp/r4_oracle.pt timeout 900 python -m pytest tests/test_parity_r4_gpu.py -m gpu -q -k "configs" 2>&1 | tail -5 | tee -a $L
cp gpurun_out/parity.jsonl gpurun_out/r4_run3_parity.jsonl
timeout 600 python tools/r4/parity_variants.py /tmp/r4_oracle.pt default tail_hilo tail_hilo+branch_f32 > gpurun_out/r4_run3_parity_variants.jsonl 2> gpurun_out/r4_run3_parity_variants.err
cat gpurun_out/r4_run3_parity_variants.jsonl | tee -a $L; tail -2 gpurun_out/r4_run3_parity_variants.err
timeout 400 python bench.py --steps 2 --no-cpu-baseline > gpurun_out/r4_run3_bench_default_with_throughput_mode.json 2> gpurun_out/r4_run3_bench.err
python - <<'PY' | tee -a gpurun_out/r4_run3.log
import json
d = json.load(open('gpurun_out/r4_run3_bench_default_with_throughput_mode.json'))
print('default line', round(d['value'], 4), round(d['ms_per_step'], 1), 'throughput_mode', d.get('throughput_mode'))
for l in open('gpurun_out/r4_run3_parity.jsonl'):
    e = json.loads(l); e.pop('latents_rel_l2_per_step', None); print(e)
PY
