"""Round-6 documentation filler: writes the round-6 blocks of DESIGN.md from one table of measured numbers (the placeholders @@ROUND6_*@@
are replaced in place; run once — kept for the record of where every number in those paragraphs comes from)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LEDGER_TAIL = '''| 13 | the whole block — attn1 → attn2 → attn_temporal → ff — in one launch (`r06_whole_block_launch_vs_attention_plus_feed_forward_launches_run13.jsonl`, `r06_bench1_run13_ab_whole_block_{on,on_repeat,off}.json`) | 4.51 ms against 2.24 + 2.61 at M = 409 600, 1.37 against 0.69 + 0.79 at 102 400; same-box A/B **1.2205 → 1.2307 / 1.2284 (+0.75 %)**.  In its first build hipcc kept the lane address of `__shfl_xor`'s `ds_bpermute` across the temporal head loop by parking it in `a0` — a NAMED accumulator, i.e. live data — and the build audit refused the library; the half-wave reductions of these kernels now use `v_permlane32_swap` (no address register).  The file was split into shared headers + one translation unit per heavy instance: build 5.5 → 2.5 min |
| 14, 15 | GroupNorm apply → proj_in inside that launch (`r06_bench1_run14_ab_proj_in_fused_{on,on_repeat,off}.json`, `r06_tests_gpu_suite_and_smoke_run15.log`, `r06_parity_full_suite_run15_transformer_launch.jsonl`) | the GroupNorm-apply pass (0.24 ms) and the proj_in launch (0.53 ms, memory-bound) of every 512-channel transformer disappear for one read of the GroupNorm input and 512 MFMAs per wave in the launch's prologue: same-box A/B **1.2135 → 1.2302 / 1.2263 (+1.2 %)**; 225 GPU tests + smoke; headline 7.30e-4 / 8.70e-4, configs[2] 6.65e-4 / 8.71e-4 (unchanged to 1e-5) |
| 16 | **evidence of the final tree** on one (slow) box (`r06_final_run16_*`): default and driver-style lines, rocprofv3 kernel table, PMC traffic, per-shape table, configs[2] / [3] / [4] lines, two clips per GPU, T = 32 parity | driver-style (`--steps 20 --warmup 2`) **1.192 frames/s** (6 711 ms per clip; the boxes of runs 13 / 14 gave 1.23 on the same tree), conv 906 TFLOP/s = **0.362** over 6 345 launches (0.375 on run 14's box), HBM traffic 1.23 GB per conv launch = 1.36 × algorithmic; the transformer launch 864 TFLOP/s (0.35 of the MFMA peak, 15 % of GPU time); two clips per GPU 1.25; configs[2] 1.196, configs[3] (T = 32, one GPU) 1.003, configs[4] tile with the video VAE 1.171; rocprofv3 and the HIP events agree (conv_gemm256w<2,0,0>: 1 009 µs average over 8 700 launches); T = 32 parity 7.0e-4 / 8.5e-4 again |
'''

N = dict(
    fps_driver="1.192", fps_default="1.193", ms_clip="6 711", two_clips="1.25", conv_frac="0.362", conv_tflops="906", n_tests="225",
    traffic="1.23 GB", traffic_ratio="1.36", block_ms="5.2 / 1.5", n_kernels="98", lib_mb="3.3", conv_launches="6 345", ledger_tail=LEDGER_TAIL,
)


def main():
    N.update(dict(a.split("=", 1) for a in sys.argv[1:]))
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    summary = f"""**Round 6 in one paragraph** (details in §3, §4, §6; rounds 1-5: `profiles/HISTORY.md`).  The round-5 verdict: parity green on the headline by 5 %
on one metric only, throughput back where round 2 had it (driver: 1.105 → 1.019 → 0.996 → 1.104 frames/s), and what is left is "bytes and idle tile
phases, not the k-step": the K = 512 linears at 0.21 of the MFMA peak, every transformer sub-layer round-tripping the fp32 stream 3–4 times.  This
round (i) **closed the parity evidence**: the tiled video-VAE fixture at its own 30-step schedule (`pipe_tiled_full_videovae_30`: 1.12e-3 — outside 1e-3 on
3-frame 68 × 160 tiles and reported as such), a direct fp32 `F.conv3d` / GEMM reference for every case of the four-wave conv kernel, configs[3] at its
real length (T = 32, 320×320, 30 steps vs the GPU oracle: latents 7.0e-4, `.images` 8.5e-4; windows on two streams bit-identical to serial), a
full-width forward at the stated 1e-3 inside `smoke()` (6.2e-4); (ii) **bought the margin**: block tails leave their producer as the hi | lo operand
pair (no cast pass) and that mode is the default — headline latents 8.2e-4 → **7.3e-4**, `.images` 9.5e-4 → **8.7e-4** over all pixels / 1.23e-3 →
1.12e-3 unclamped, the two headline tests now assert 8.0e-4 / 9.2e-4; (iii) built the **fused transformer kernels** the verdict named and went on
from there (`csrc/xattn_common.h`, `xattn_fused.hip`, `tattn_*.hip`): every sub-layer of the 512-channel `BasicTransformerBlock`s — LayerNorm →
projection(s) → attention → `to_out` → + residual, and LayerNorm → GEGLU → down → + residual — with lane = token from the first load to the last store,
weights streamed as pre-packed MFMA fragments through an LDS ring, the fp32 stream itself in the accumulator file; then two of them, three of them,
the whole block, and finally **GroupNorm apply → proj_in → attn1 → attn2 → attn_temporal → ff of a `Transformer3DModel` in ONE launch** that reads the
stream once and writes only the hi | lo operand pair of `proj_out`: per clip 300 launches of {N['block_ms']} ms in place of 5 100 (LayerNorm 1 200, projections
2 100, attention 900, GroupNorm apply 300, feed-forward GEMMs 600), the K ≤ 1 024 linear class and the GEGLU class of round 5's table are gone
at these levels; (iv) took the HIP events out of the timed region (same-box A/B: +0.63 %; `roofline` and the per-kernel table come from one
instrumented clip behind it); (v) **pruned the library**: one translation unit per kernel family, the legacy / ablation / trace instances behind
`-DUAV_DEV_KERNELS` in a side library, 115 → {N['n_kernels']} kernels, 5.4 → {N['lib_mb']} MB, and build audits that fail on ANY scratch in a shipped kernel and on any
compiler-generated use of the accumulator file inside the kernels that name it (it caught hipcc parking a live value in a named accumulator);
(vi) measured and recorded what lost: the two guidance branches of one clip on two streams (−3.4 %, again).  Serial headline, final tree, the way
the driver runs it: **{N['fps_driver']} frames/s** on a slow box, 1.23 on the boxes of runs 13 / 14 ({N['ms_clip']} ms per clip; round 5: 1.117–1.124; two clips per GPU {N['two_clips']}), conv **{N['conv_tflops']} TFLOP/s =
{N['conv_frac']}** of the dense peak over its {N['conv_launches']} launches (traffic {N['traffic_ratio']}× algorithmic), {N['n_tests']} GPU tests + smoke green.
"""
    s = s.replace("@@ROUND6_SUMMARY@@", summary)
    ledger = f"""### Round 6 — measurements in the order they were taken (one `gpurun` call each, `tools/run.sh`; boxes differ by ±3 %)

| run | what | result |
|---|---|---|
| 1 | first version of the fused cross-attention sub-layer kernel (weights as LDS-DMA bursts behind each barrier), full suite, `--overlap-split-cfg` (`r06_xattn_fused_vs_four_launch_chain_run1_first_version.jsonl`, `r06_bench1_run1_*.json`, `r06_bench1_cfg_branches_on_two_streams_run1.json`) | every kernel test green on the first run; 0.98 ms against 1.27 ms (chain) at M = 409 600, 0.31 / 0.37 at 102 400; clip 1.103 → **1.122** frames/s same box; the two guidance branches of ONE clip on two streams: **1.084 (−3.4 %)** — half-size launches and the CFG-shared head given up cost more than the overlap gives: recorded, not pursued (VERDICT r5 next #6) |
| 2 | DMA pieces between the MFMAs + first MFMA on C = 0 | NaN: the 12-bit instruction offset of `buffer_load … lds` moves BOTH the global and the LDS address (the pieces landed 1–3 KiB too far) — fixed in run 3 |
| 3, 4 | phase trace of the kernel (`tools/trace_xattn.py`, the `-DUAV_DEV_KERNELS` side library; `r06_xattn_fused_phase_trace_run3/4.jsonl`), bigger load batches, row-coalesced stores through the idle ring | per 128-token tile (ticks): first read of x (LayerNorm statistics) 30 k, second read (operands + accumulators) 20 k, head 0 (cold ring) 8.7 k, then **7.9 k per head** — Q GEMM 2.5 k (64 MFMA: 39 cycles each), S 0.8 k, softmax 1.7 k, PV 0.8 k, Wout 2.9 k — and 13 k for the stores: 129 k per tile, of which the 8 heads' 65 k are 60 % MFMA time.  0.98 → 0.87 ms; event-overhead A/B twice interleaved: 1.1420 / 1.1414 with conv events in the timed region, **1.1493 / 1.1486 without (+0.63 %)** (`r06_event_overhead_ab_*_run4.jsonl`) |
| 4, 5 | block tails as hi \\| lo pairs from the producer's epilogue (`UAV_CONV_OUT_HILO`), default on; T = 32 parity (`r06_parity_headline_tail_hilo_on_run4.jsonl`, `r06_parity_configs3_t32_320x320_30steps_vs_gpu_oracle_run5.jsonl`, `r06_parity_full_suite_run5_*.jsonl`, `r06_bench1_run5_*.json`) | bit-identical to fp32 result + cast pass (after keeping hipcc from contracting `v·scale − hi` into one fma); headline 8.2e-4 / 9.5e-4 / 1.23e-3 → **7.3e-4 / 8.7e-4 / 1.12e-3**, full-width forward 8.5e-4 → 6.5e-4, configs[0] 1.55e-3 → 1.25e-3, tiled 30-step fixture 1.25e-3 → 1.12e-3; 1.1484 → 1.1352 frames/s (−1.2 %) same box; **T = 32 at 320²: latents 7.0e-4, `.images` 8.5e-4 / 1.09e-3**, 35.3 s serial, 34.2 s with the windows on two streams (bit-identical) |
| 6 | two cross-attention sub-layers in one launch: accumulators by NAME in the accumulator file, second LayerNorm on them (`r06_bench1_run6_cross_pair_fused.json`) | pair 1.40 ms against 2 × 0.90 single / 2 × 1.22 chain at M = 409 600; clip **1.158**; as C++ tuples that asm statements hold as `"+a"` AND the VALU reads, the accumulators cost 34 … 1 679 spilled registers per lane — by name, 0 |
| 7 | temporal sub-layer kernel (`r06_fused_sublayers_vs_chains_run7_cross_pair_and_temporal.jsonl`, `r06_bench1_run7_*.json`, `r06_tests_gpu_suite_and_smoke_run7.log`) | all four cases green on the first run; **1.14 ms against 1.85 ms** (LayerNorm 0.23 + q\\|k\\|v 0.83 + attention 0.34 + to_out 0.45) at M = 409 600, 0.35 / 0.53 at 102 400; clip **1.193 frames/s**, conv 0.371; 214 GPU tests + smoke |
| 8 | block kernel: attn1 → attn2 → attn_temporal of a block in one launch (`r06_fused_sublayers_vs_chains_run8_block_of_three.jsonl`, `r06_bench1_run8_block_kernel.json`) | 2.22 ms against 1.42 (pair) + 1.13 (temporal) at M = 409 600; 469 ms per clip in this class against 521; 215 GPU tests + smoke.  131 spilled registers in the first form — the temporal loop's fragments went into other registers than the cross loops' and the move between the two sets went through scratch; a second fragment array (the third LayerNorm writes `xt`, not `xn`): 0 |
| 9, 10 | norm3 (the LayerNorm in front of the feed-forward) written by that launch's epilogue; per-shape table; PMC traffic of the conv kernels re-taken for the new sources (`r06_bench1_run9_*.json`, `r06_run10_per_shape_table_before_fused_feed_forward.txt`) | 300 LayerNorm launches per clip gone, clip **1.190**; where the clip goes now: GroupNorm apply 10.8 %, the 3×3 convs at 1.15–1.25 PFLOP/s, the block kernel 722 TFLOP/s, GEGLU 512 → 4 096 still 704 TFLOP/s (2.44 ms a launch) and 2 048 → 512 890: the feed-forward is the worst class left |
| 11, 12 | feed-forward sub-layer kernel (`r06_fused_feed_forward_vs_three_launch_chain_run11.jsonl`, `r06_bench1_run12_ab_feed_forward_fused_{{on,on_repeat,off}}.json`, `r06_tests_gpu_suite_and_smoke_run12.log`, `r06_parity_full_suite_run12_fused_feed_forward.jsonl`) | every kernel test green on the first run; **2.80 ms against 3.60** (LayerNorm 0.22 + GEGLU GEMM 2.46 + down GEMM 0.98) at M = 409 600 = 921 against 717 TFLOP/s, 0.84 / 0.96 at 102 400 (800 tiles on 256 CUs: 3.1 rounds); same-box A/B **1.172 → 1.201 / 1.202 frames/s (+2.5 %)**; 221 GPU tests + smoke; headline parity unchanged (latents 7.30e-4, `.images` 8.69e-4; configs[2] 6.70e-4 / 8.76e-4).  The first form (slices of 64: value and gate as two "Q" steps, 64 fp32 results live) spilled 4 registers and, with the audit compiling the file to assembly once per KERNEL, took 12 minutes to build; slices of 32 with value \| gate as the two channel tiles of one step: 0 spills, and the audit now compiles each file once, beside the object compiles |
{N['ledger_tail']}"""
    s = s.replace("@@ROUND6_LEDGER@@", ledger)
    open(p, "w").write(s)


if __name__ == "__main__":
    main()
