#!/bin/bash
# round 3, GPU call 6: BASELINE configs[4] as ONE WHOLE CLIP on one GPU (32 x 540x960 -> 2160x3840, 8 tiles, vae_video, Wavelet fix)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/bench_config5.py > gpurun_out/r3_bench_config5_whole_clip.json 2> gpurun_out/r3_bench_config5_whole_clip.err
tail -3 gpurun_out/r3_bench_config5_whole_clip.err
cat gpurun_out/r3_bench_config5_whole_clip.json
