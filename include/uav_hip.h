/*
 * uav_hip.h — C ABI of libuav_hip.so, the MI355X (gfx950) kernel library behind the
 * Upscale-A-Video hot path (DDIM loop over UNetVideoModel + video-VAE decode + RAFT /
 * flow-guided latent propagation).
 *
 * The reference (sczhou/Upscale-A-Video) has NO native/FFI layer: its hot path is ATen ops
 * issued from Python.  The drop-in boundary is therefore the Python object
 * protocol of `models_video.*`; those classes live in
 * `upscale-a-video_amd/models_video/` and call the entry points below through ctypes
 * (`upscale-a-video_amd/uav/_lib.py`).  Each entry point names the reference op(s) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the call (except the params structs
 *     themselves, which are host memory read during the call);
 *   - activations are channels-last: a video tensor (B,C,T,H,W) of the reference is held as
 *     rows [B*T*H*W][C] (fp16), "image" index = b*T + t;
 *   - all launches are stream-ordered on `stream` (a hipStream_t passed as void*), the library
 *     keeps no global state and never allocates; workspaces are caller supplied;
 *   - return value: 0 on success, negative UAV_E* on invalid arguments, positive = hipError_t
 *     of a failed launch.  No exceptions cross the ABI.
 */
#ifndef UAV_HIP_H
#define UAV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAV_ABI_VERSION 6

#define UAV_EINVAL   (-1)   /* bad argument (null pointer, size not supported) */
#define UAV_EALIGN   (-2)   /* pointer / stride alignment requirement violated */
#define UAV_ESHAPE   (-3)   /* shape constraint of the kernel violated */

/* ---- library info ------------------------------------------------------------------- */
int  uav_version(void);
/* 1 in a development build (-DUAV_DEV_KERNELS, tools/ab/build_dev.sh: ablation / trace / legacy kernel instances and the switches that
 * reach them — UAV_CONV_DBG / _PERSIST / _DMAV / _SK, UAV_CONV_W4_TRACE, UAV_LN_FOLD, UAV_ATTN512=0), 0 in the product library */
int  uav_has_dev_kernels(void);
/* 0 if device `dev` is a gfx950 (MI355X); UAV_EINVAL otherwise. `name_out` (optional, >=64 B). */
int  uav_device_check(int dev, char* name_out);

/* ---- K1/K2: implicit-GEMM convolution / linear on MFMA (fp16 in, fp32 accumulate) -----
 * Replaces: InflatedConv3d = per-frame nn.Conv2d (resnet.py:94-101), nn.Conv3d (k,1,1) and
 * (3,3,3) (resnet.py:332,348,461), Upsample3D/Downsample3D convs (resnet.py:104-197),
 * every nn.Linear of the transformer blocks (attention.py:97-106,327,355; GEGLU
 * diffusers_attention.py:801-823), with fused epilogues: bias, per-batch time-embedding add
 * (resnet.py:272-276), residual add, 1/output_scale_factor (resnet.py:292), GEGLU gate,
 * skip-concat read from two tensors (unet_blocks.py:563), nearest-2x upsample folded into
 * the gather (resnet.py:144).
 *
 *   out[m][n] = scale * ( sum_{tap,c} A[src(m,tap)][c] * W[n][tap*cin + c]
 *                         + bias[n] + rowbias[m / rows_per_batch][n] + residual[m][n] )
 *
 * A is [n_img*hi*wi][c1] (+ optional second source [..][c2], channel-concatenated).
 * W is the PACKED weight: fp16 [n_pad][k_pad], k = tap*cin_p + c, rows >= n and k >= K zero.
 */
#define UAV_CONV_GEGLU    1u   /* rows of W interleaved [32 value | 32 gate]; out width n/2 */
#define UAV_CONV_OUT_F32  2u   /* store fp32 instead of fp16 */
#define UAV_CONV_GELU       256u /* out = gelu_erf(conv + bias + rowbias) (+ residual) * scale  (CLIP ViT-H MLP) */
#define UAV_CONV_QUICK_GELU 512u /* x * sigmoid(1.702 x) (OpenAI CLIP MLP) */
#define UAV_CONV_RES_F32  128u /* `residual` is fp32 [M][res_stride] (fp32 residual stream of the VAE decoder) */
/* Round 5's short-K kernel (1x1 / stride-1 launches with K = c1 + c2 <= 1024 and n a multiple of 256: 128 x 256 tile, two workgroups per
 * CU) is a DEVELOPMENT kernel since round 6: it is compiled only into tools/ab/libuav_hip_dev.so (-DUAV_DEV_KERNELS) and even there off
 * unless UAV_CONV_SK=1 — the product library runs these launches in the four-wave kernel (and the 512-channel transformer sub-layers in
 * the fused kernels below).  The flag keeps a launch out of the short-K kernel where that kernel exists; the product ignores it. */
#define UAV_CONV_NO_SHORTK 1024u
/* Round 5: launches of the 256x256-tile class run in the four-wave kernel (conv_gemm256w_kernel: one wave per SIMD, 128 x 128 wave
 * tiles, accumulators in the accumulator file) — same results bit for bit as the 8-wave kernel.  This flag keeps a launch in the
 * 8-wave kernel (A/B measurements and the bit-identity tests). */
#define UAV_CONV_NO_W4 2048u
/* Round 6: with UAV_CONV_OUT_F32, store every fp32 result v as the fp16 pair hi = fp16(v), lo = fp16(v - hi) instead: `out` is
 * fp16 [M][out_stride >= 2 n], hi in columns 0 .. n-1, lo in n .. 2n-1 — the operand rows of a 1x1 consumer whose weights are
 * repeated along K (block tails: the last feed-forward of a Transformer3DModel -> proj_out, attention.py:389-398; the tail ResNet of
 * a TemporalModule3D -> shift_conv, temporal_module.py:175-194).  Bit-identical to the fp32 result followed by
 * uav_cast_f32_hilo, without the pass.  Only launches uav_conv_gemm_hilo_ok() accepts. */
#define UAV_CONV_OUT_HILO 4096u

typedef struct {
    const void*  a1;            /* fp16 source 1, rows of c1 channels */
    const void*  a2;            /* fp16 source 2 (channels c1..c1+c2) or NULL */
    int32_t      c1, c2;        /* c1+c2 = cin; both multiples of 64, or (c1==8,c2==0) "small" mode */
    const void*  w;             /* packed fp16 weights [n_pad][k_pad] */
    const float* bias;          /* fp32 [n_pad] or NULL */
    const float* rowbias;       /* fp32 [n_batches][rowbias_stride] or NULL */
    int32_t      rows_per_batch;
    int32_t      rowbias_stride;
    const void*  residual;      /* fp16 (fp32 with UAV_CONV_RES_F32) [M][res_stride] or NULL */
    int32_t      res_stride;
    void*        out;           /* fp16 (or fp32) [M][out_stride] */
    int32_t      out_stride;
    int32_t      n_img, t_len;  /* images (= batch*frames), frames per batch element */
    int32_t      hi, wi;        /* input spatial size */
    int32_t      ho, wo;        /* output spatial size */
    int32_t      kt, kh, kw;    /* taps */
    int32_t      stride;        /* spatial stride (1 or 2) */
    int32_t      pad_t, pad_h, pad_w;
    int32_t      upsample;      /* 1: input is nearest-2x upsampled on the fly (ho=2*hi) */
    int32_t      n;             /* logical output channels (GEGLU: 2*features) */
    int32_t      n_pad, k_pad;  /* packed weight dims: n_pad%128==0, k_pad%64==0 */
    float        out_scale;
    uint32_t     flags;
    const void*  zero_page;     /* >=256 B of device zeros (padding source for the DMA gather) */
    /* Fused GroupNorm statistics for the CONSUMER of `out` (resnet.py:267-284: conv -> GroupNorm -> SiLU -> conv): with
     * gn_partials != NULL the epilogue also writes, for every block of `rows` = uav_conv_gemm_gn_chunk_rows() output rows
     * (chunk = m / rows) and every group g of n / gn_groups channels, the fp32 sum and sum of squares of the values it
     * stores: gn_partials[(which*gn_groups + g) * (M/rows) + chunk], which = 0 sum, 1 sum of squares (2*gn_groups*M/rows
     * floats, every element written).  uav_groupnorm_finalize_partials turns them into scale/shift tables, which saves
     * the statistics pass over `out`.  NULL / 0: off (uav_conv_gemm_f32 ignores both). */
    void*        gn_partials;
    int32_t      gn_groups;
    /* Strided output rows (the four sub-pixel phases of "nearest 2x, then 3x3 conv", resnet.py:144-158, each a 2x2 conv on
     * the low-resolution input — 16 instead of 36 multiply-adds per output pixel and channel pair): with out_map_w > 0
     * output row m = Y*out_map_w + x (x < out_map_w) is stored at row Y*out_map_sy + x*out_map_sx + out_map_off of `out`
     * instead of row m.  residual must be NULL, no GEGLU; gn_partials only with gn_chunk_cpi (below).  0: rows in order. */
    int32_t      out_map_w, out_map_sy, out_map_sx, out_map_off;
    /* Source 2 read batch-broadcast: a2 holds a2_images = n_img/2 images and images a2_images .. n_img-1 read the pixels
     * of images 0 .. a2_images-1 (the skip tensors of the CFG-shared UNet head exist once, unet_blocks.py:563 concatenates
     * them to both batch entries).  0: a2 has n_img images.  stride 1, no upsampling. */
    int32_t      a2_images;
    /* Source 2 takes part in the CENTRE tap only (a 1x1 conv of a2 summed into a kh x kw conv of a1 in one accumulator: the
     * shortcut conv of a ResNet block folded into its conv2, resnet.py:286-292 — `conv_shortcut(x) + conv2(h)`): the packed
     * weights keep the layout k = tap*(c1+c2) + c with ZERO entries for (tap != centre, c >= c1), so every kernel computes
     * the same sum; the 256x256 kernel skips those k-steps.  Needs stride 1, kt == 1, pad = k/2, no upsampling.  0: off. */
    int32_t      a2_center_tap;
    /* LayerNorm folded into the consuming projection (attention.py:523-564: `attn(norm(x))`, `ff(norm3(x))` — LayerNorm, then
     * a Linear): LN(x).W^T + b = rstd_m * (x . (W o gamma)^T - mu_m * colsum(W o gamma)) + (W.beta + b).
     *   PRODUCER of x (a linear with an fp32 result): ln_raw_out != NULL -> also writes fp16(x) rows [M][n] (the consumer's MFMA
     *     operand) and, per row and 128-column chunk c, (sum, sum of squares) of the fp32 values: ln_stat_out[(c*M + m)*2 + {0,1}],
     *     n/128 chunks.
     *   CONSUMER (1x1, fp16 result, no residual / time embedding; GEGLU allowed): ln_stat_in != NULL -> a1 = fp16(x) rows, w = the
     *     packed fp16(W o gamma), bias = W.beta + b (fp32), ln_colsum[n_pad] = row sums of the packed weights (fp32), ln_chunks /
     *     ln_n / ln_eps = chunk count, LayerNorm width (= c1) and epsilon.
     * Only launches for which uav_conv_gemm_ln_ok() returns 1 (256x256 kernel, full wave tiles); else UAV_ESHAPE.  NULL: off. */
    void*        ln_raw_out;
    float*       ln_stat_out;
    const float* ln_stat_in;
    const float* ln_colsum;
    int32_t      ln_chunks, ln_n;
    float        ln_eps;
    /* ABI v5.  GroupNorm partials of SEVERAL launches in one workspace (the four sub-pixel phase convs of an up-sampler write
     * interleaved rows of ONE output tensor, resnet.py:144-158): with gn_chunk_cpi > 0 the launch's chunk index k = m / rows is
     * stored at (k / gn_chunk_cpi) * gn_chunk_stride + gn_chunk_off + k % gn_chunk_cpi of a workspace that holds
     * gn_chunks_total chunks per (statistic, group) — e.g. cpi = chunks per frame of this launch, stride = 4 * cpi, off = phase *
     * cpi: every frame of the output then owns one contiguous run of chunks and uav_groupnorm_finalize_partials reads the
     * workspace as if one launch had written it.  Lifts the "no out_map" rule of gn_partials.  0: chunk k is stored at k. */
    int32_t      gn_chunk_cpi, gn_chunk_stride, gn_chunk_off;
    int64_t      gn_chunks_total;
} uav_conv_params;

int uav_conv_gemm_f16(const uav_conv_params* p, void* stream);
/* 1 if uav_conv_gemm_f16(p) can honour p's LayerNorm-fold fields (ln_raw_out / ln_stat_in), else 0.  Host-only. */
int uav_conv_gemm_ln_ok(const uav_conv_params* p);
/* Rows per statistics chunk (64) if uav_conv_gemm_f16(p) can produce GroupNorm partials for p->gn_groups, else 0 (small
 * launches, N tails, GEGLU / activation epilogues, groups of other than 4..128 channels): decide BEFORE setting
 * gn_partials — a launch that cannot honour the request returns UAV_ESHAPE.  Host-only, no device work. */
int uav_conv_gemm_gn_chunk_rows(const uav_conv_params* p);
/* 1 if this launch can store its result as the hi | lo pair (UAV_CONV_OUT_HILO): four-wave kernel, every wave tile on the
 * row-coalesced fp32 epilogue with an fp32 residual (bias, M % 64 == 0, n % 128 == 0, no time-embedding row, no statistics) */
int uav_conv_gemm_hilo_ok(const uav_conv_params* p);

/* ---- K3: GroupNorm statistics + apply (+SiLU) ---------------------------------------
 * Replaces nn.GroupNorm on 5-D tensors (statistics over C/G x T x H x W: resnet.py:267,278,
 * 366,377,467,478,495; unet_video.py:567) and on per-frame 4-D tensors (attention.py:374;
 * unet_blocks.py:740) followed by SiLU (resnet.py:268,284).
 *   x: fp16 (or fp32, x_f32 = 1) rows [n_inst*rows_per_inst][c] read from up to two channel-concatenated sources.
 *   stats pass : partial per-channel (sum, sumsq) -> scale/shift tables [n_inst][c] (fp32):
 *                scale = gamma*rstd, shift = beta - mean*rstd*gamma.
 *   apply pass : y = act(x*scale + shift), fp16, written contiguously [rows][c1+c2].
 */
int64_t uav_groupnorm_workspace_bytes(int32_t n_inst, int32_t c);
int uav_groupnorm_scale_shift(const void* x1, const void* x2,
                              int32_t x_f32, /* 0: x rows are fp16, 1: fp32 (fp32 residual stream of the VAE decoder) */
                              int32_t c1, int32_t c2,
                              int64_t x2_rows, /* 0: x2 has as many rows as x1; else x2 has x2_rows = half of them and is
                                                  read batch-broadcast (row r >= x2_rows reads row r - x2_rows) */
                              int32_t c_real, /* real channels (<= c1+c2); padding gets scale=shift=0 */
                              int32_t n_inst, int64_t rows_per_inst, int32_t groups, float eps,
                              const float* gamma, const float* beta,
                              float* scale_out, float* shift_out,
                              void* workspace, int64_t workspace_bytes, void* stream);
/* scale/shift tables from the partials a conv epilogue wrote (uav_conv_params.gn_partials): `partials` is
 * [2][groups][chunks_total] fp32, an instance owns rows_per_inst / chunk_rows consecutive chunks
 * (rows_per_inst % chunk_rows == 0, n_inst * rows_per_inst / chunk_rows == chunks_total). */
int uav_groupnorm_finalize_partials(const float* partials, int64_t chunks_total, int32_t chunk_rows, int32_t c,
                                    int32_t n_inst, int64_t rows_per_inst, int32_t groups, float eps,
                                    const float* gamma, const float* beta, float* scale_out, float* shift_out, void* stream);
/* same for a channel-concatenated input [x1 | x2] whose two sources each carry the partials of their own producer
 * ([2][groupsS][chunks_totalS]): the output groups must not straddle the seam (c1 % (c/groups) == 0) and must be whole
 * multiples of each producer's group; n_instS = instances the source holds (n_inst, or a divisor of it when the source
 * exists once for several batch entries: instance i reads the chunks of instance i % n_instS). */
int uav_groupnorm_finalize_partials2(const float* partials1, int64_t chunks_total1, int32_t c1, int32_t groups1, int32_t n_inst1,
                                     const float* partials2, int64_t chunks_total2, int32_t c2, int32_t groups2, int32_t n_inst2,
                                     int32_t chunk_rows, int32_t n_inst, int64_t rows_per_inst, int32_t groups, float eps,
                                     const float* gamma, const float* beta, float* scale_out, float* shift_out, void* stream);
int uav_groupnorm_apply(const void* x1, const void* x2, int32_t x_f32, int32_t c1, int32_t c2, int64_t x2_rows,
                        int32_t n_inst, int64_t rows_per_inst,
                        const float* scale, const float* shift, int32_t silu,
                        void* y, void* raw_f16_out /* optional: the un-normalised [x1|x2] rows as fp16 (operand of a 1x1
                        shortcut conv reading an fp32 stream); NULL = not written */,
                        int32_t raw_hilo /* 0: rows of c values fp16(x); 1: rows of 2c values [fp16(x) | fp16(x - fp16(x))] */,
                        void* stream);

/* ---- LayerNorm over the channel axis (nn.LayerNorm(dim), eps 1e-5: attention.py:457-494) */
int uav_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta,
                      int64_t rows, int32_t c, float eps, void* stream);
/* same, fp32 input rows (the UNet's fp32 residual stream, UNetVideoModel.stream_dtype); fp16 output (an MFMA operand) */
int uav_layernorm_f32in(const float* x, void* y, const float* gamma, const float* beta,
                        int64_t rows, int32_t c, float eps, void* stream);

/* ---- K5/K6/K8: flash attention on MFMA ------------------------------------------------
 * Replaces CrossAttention._attention (attention.py:209-238: baddbmm + softmax + bmm) for
 * spatial self-attention and text cross-attention, and the VAE AttentionBlock core
 * (diffusers_attention.py:341-369).  softmax(scale * Q K^T) V, fp32 softmax, no mask.
 *   q: [bq][lq] rows, element (b,i,h,d) at q[(b*lq+i)*q_stride + h*head_dim + d]
 *   k,v: [bk][lk] rows likewise; query batch b reads kv batch b / q_per_kv.
 *   head_dim in {64,128,512}.
 */
int uav_attention_f16(const void* q, int64_t q_stride, const void* k, int64_t k_stride,
                      const void* v, int64_t v_stride, void* out, int64_t o_stride,
                      int32_t bq, int32_t lq, int32_t lk, int32_t q_per_kv,
                      int32_t heads, int32_t head_dim, float scale,
                      int32_t causal /* 1: key j visible to query i only if j <= i (CLIP text encoder; lq == lk, d 64|128) */,
                      const void* zero_page /* >=16 B of device zeros */, void* stream);

/* ---- K5b (round 6): fused text cross-attention SUB-LAYER of BasicTransformerBlock ----------
 * Replaces, for the 512-channel levels of the UNet (8 heads x 64, <= 96 text keys), the four launches of one
 * `hidden = attn(norm(hidden), encoder_hidden_states) + hidden` step (attention.py:523-564 steps attn1 with
 * only_cross_attention / attn2; CrossAttention.forward :177-238 = to_q, _attention :209-238, to_out):
 *
 *     out[m][:] = x[m][:] + b_out + W_out . softmax(scale * (W_q . LayerNorm(x[m][:])) . K_b^T) . V_b ,   b = m / rows_per_kv
 * (applied once or twice: see n_subs)
 *
 * x / out: fp32 token-stream rows [rows][channels] (out may alias x: a workgroup reads its 128 rows before it writes them);
 * LayerNorm, Q, P and the attention output are rounded to fp16 exactly where the unfused chain stores them.
 *   wq_packed / wo_packed: the to_q / to_out weights [channels][channels] fp16 in MFMA-fragment stream order, kv_packed: the
 *   text K | V of every kv batch in the same order — uav_xattn_pack_kv (activations) and uav.ops.pack_xattn_weight (weights)
 *   produce them; csrc/xattn_fused.hip states the order.
 *   rows_per_kv % 128 == 0, rows % rows_per_kv == 0, channels == 512, heads == 8, lk <= 96; else UAV_ESHAPE (the caller
 *   keeps the four-launch chain for such shapes). */
typedef struct {
    const float* ln_gamma;      /* LayerNorm in front of the sub-layer: fp32 [channels] */
    const float* ln_beta;
    float        ln_eps;
    const void*  wq_packed;     /* to_q weights, fragment stream (uav.ops.pack_xattn_weight 'q') */
    const void*  kv_packed;     /* text K | V of this sub-layer's to_k / to_v (uav_xattn_pack_kv) */
    const void*  wo_packed;     /* to_out weights, fragment stream ('out') */
    const float* out_bias;      /* to_out bias: fp32 [channels] */
} uav_xattn_params;
/* n_subs = 1: one sub-layer; n_subs = 2: attn1 (only_cross_attention) and attn2 of one block back to back on the same rows — the
 * first one's output stays in the accumulators, the second LayerNorm runs on it in registers (one read and one write of the stream
 * for both). */
int uav_xattn_sublayers_f32(const float* x, float* out, const uav_xattn_params* subs, int32_t n_subs, int64_t rows,
                            int32_t rows_per_kv, int32_t lk, int32_t channels, int32_t heads, float scale, void* stream);
/* ---- K7b (round 6): fused TEMPORAL attention SUB-LAYER of BasicTransformerBlock ------------------------------------------------
 * Replaces the four launches of `hidden = attn_temporal(norm_temporal(hidden)) + hidden` (attention.py:555-560; TemporalAttention
 * :626-733 = to_q | to_k | to_v, RoPE + RelativePositionBias + per-pixel softmax over the T frames, to_out) on fp32 stream rows
 * [n_batch * t_len * hw][channels] (row (b T + t) hw + pixel): out = x + b_out + W_out . attention, the stream read and written once.
 * channels == 512, heads == 8, t_len == 8, rot_dim == 32, hw % 16 == 0; else UAV_ESHAPE (the caller keeps the chain).  out may alias x
 * (a workgroup's 16 pixels x 8 frames are disjoint from every other workgroup's rows and are read before they are written). */
typedef struct {
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void*  wq_packed;     /* to_q / to_k / to_v weights, each in the 'q' fragment order of uav.ops.pack_xattn_weight */
    const void*  wk_packed;
    const void*  wv_packed;
    const void*  wo_packed;     /* to_out weights ('out' order) */
    const float* out_bias;
    const float* rel_bias;      /* fp32 [heads][t_len][t_len] */
    const float* rope_cos;      /* fp32 [t_len][rot_dim / 2] */
    const float* rope_sin;
    int32_t      rot_dim;
    /* optional (next_ln_out != NULL): the LayerNorm that FOLLOWS the sub-layer in the block (norm3, in front of the feed-forward) applied
     * to the rows written to `out`, as fp16 operand rows [rows][channels] — saves that LayerNorm launch (attention.py:562-564) */
    void*        next_ln_out;
    const float* next_ln_gamma;
    const float* next_ln_beta;
    float        next_ln_eps;
} uav_tattn_params;
int uav_tattn_sublayer_f32(const float* x, float* out, const uav_tattn_params* p, int32_t n_batch, int32_t t_len, int64_t hw,
                           int32_t channels, int32_t heads, float scale, void* stream);
/* attn1 -> attn2 -> attn_temporal of one BasicTransformerBlock with only_cross_attention (attention.py:523-564) in ONE launch: `cross`
 * = the two text cross-attention sub-layers (uav_xattn_params[2], K | V packed per kv batch = per b of n_batch), then the temporal
 * sub-layer; the rows between the three stay in the accumulators, every LayerNorm but the first runs on them in registers.  Same shape
 * contract as uav_tattn_sublayer_f32, lk <= 96. */
int uav_block_attn_sublayers_f32(const float* x, float* out, const uav_xattn_params* cross, int32_t lk, float cross_scale,
                                 const uav_tattn_params* temporal, int32_t n_batch, int32_t t_len, int64_t hw, int32_t channels,
                                 int32_t heads, float temporal_scale, void* stream);
/* The feed-forward sub-layer of the same block (reference attention.py:562-564 `ff(norm3(x)) + x`: LayerNorm -> GEGLU 512 -> 2 x 2048
 * -> Linear 2048 -> 512 -> + residual; diffusers_attention.py:735-823) in ONE launch: the hidden activations never leave the registers.
 *   x: fp32 rows [rows][512], rows % 128 == 0; out: fp32 rows (may be NULL) and / or out_hilo: fp16 rows [rows][1024] = fp16(y) |
 *   fp16(y - fp16(y)), the operand pair of proj_out (uav_cast_f32_hilo's layout, bit-identical to it on the fp32 result).
 *   w_packed: 6 MiB fragment stream = per slice c of 32 hidden channels: 64 KiB of fragments of W_up (k-step, tile: tile 0 = value rows
 *   32c.., tile 1 = gate rows 2048 + 32c..) then 32 KiB of fragments of columns 32c.. of W_down — uav.ops.pack_ff_weights.
 *   up_bias: fp32 [4096] (value | gate), down_bias: fp32 [512].  channels must be 512, inner 2048 (UAV_ESHAPE otherwise). */
typedef struct uav_ff_params {
    const float* ln_gamma;
    const float* ln_beta;
    float        ln_eps;
    const void*  w_packed;
    const float* up_bias;
    const float* down_bias;
} uav_ff_params;
int uav_ff_sublayer_f32(const float* x, float* out, void* out_hilo, const uav_ff_params* p, int64_t rows, int32_t channels,
                        int32_t inner, void* stream);
/* The WHOLE BasicTransformerBlock (attention.py:523-564: attn1 -> attn2 -> attn_temporal -> ff, only_cross_attention) in ONE launch: the
 * stream is read once; `out` (fp32 rows, may be NULL) and / or `out_hilo` (fp16 hi | lo pair, see uav_ff_sublayer_f32) are written once.
 * Same shape contract as uav_block_attn_sublayers_f32; temporal->next_ln_out must be NULL; inner must be 2048.
 * proj_in != NULL: x is the fp32 INPUT of the Transformer3DModel's GroupNorm (attention.py:389-393) and the launch starts with GroupNorm
 * apply -> proj_in: gn_scale / gn_shift = the per-instance rows [n_batch * t_len][512] that uav_groupnorm_scale_shift /
 * uav_groupnorm_finalize_partials write (one instance per frame), w_packed = proj_in's weight as 'out'-kind fragments
 * (uav.ops.pack_xattn_weight), bias fp32 [512]. */
typedef struct uav_projin_params {
    const float* gn_scale;
    const float* gn_shift;
    const void*  w_packed;
    const float* bias;
} uav_projin_params;
int uav_block_sublayers_f32(const float* x, const uav_projin_params* proj_in, float* out, void* out_hilo, const uav_xattn_params* cross,
                            int32_t lk, float cross_scale, const uav_tattn_params* temporal, const uav_ff_params* ff, int32_t n_batch,
                            int32_t t_len, int64_t hw, int32_t channels, int32_t heads, int32_t inner, float temporal_scale, void* stream);
/* k, v: fp16 rows [n_batch * lk][stride] (head h in columns h*head_dim ..) -> out: n_batch * heads * 32 KiB */
int uav_xattn_pack_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, int32_t n_batch, int32_t lk,
                      int32_t heads, int32_t head_dim, void* out, void* stream);

/* Single-head d = 512 attention (the VAE mid block: reference vae.py AttentionBlock, softmax(Q K^T / sqrt(512)) V over the L = H W pixels of a
 * frame) on pre-packed operands: uav_attention512_pack_kv turns the K / V rows [bq * lk][stride] (fp16, 512 channels) into two MFMA-fragment
 * streams of uav_attention512_pack_bytes(bq, lk) bytes EACH (caller-allocated, 16-B aligned; [bq][ceil(lk / 32)][32 KiB]), and
 * uav_attention512_packed_f16 walks them: q, o fp16 rows [bq * lq][stride >= 512].  Same arithmetic as uav_attention_f16 at head_dim 512
 * (fp16 operands and P, fp32 scores / accumulators / online softmax with exact deferred rescale); another summation order. */
int64_t uav_attention512_pack_bytes(int32_t bq, int32_t lk);
int uav_attention512_pack_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, int32_t bq, int32_t lk, void* k_packed,
                             void* v_packed, void* stream);
int uav_attention512_packed_f16(const void* q, int64_t q_stride, const void* k_packed, const void* v_packed, void* o, int64_t o_stride,
                                int32_t bq, int32_t lq, int32_t lk, float scale, void* stream);

/* ---- K7: per-pixel temporal attention --------------------------------------------------
 * Replaces TemporalAttention._attention (attention.py:699-733): q*scale -> RoPE on the first
 * rot_dim dims of each head (interleaved pairs) -> QK^T + bias[h][i][j] -> -max -> softmax -> V,
 * for the T tokens of each (batch, pixel), reading the channels-last tensor in place
 * (token (b,t,p) at row (b*t_len+t)*hw + p), no (b f) d c <-> (b d) f c transposes.
 *   qkv: fp16 rows of 3*c (q | k | v), out: fp16 rows of c.  t_len <= 8 (the pipeline's window).
 *   rope_cos/sin: fp32 [t_len][rot_dim/2]; bias: fp32 [heads][t_len][t_len].
 */
int uav_temporal_attention_f16(const void* qkv, void* out, int32_t n_batch, int32_t t_len,
                               int64_t hw, int32_t c, int32_t heads, float scale,
                               const float* rope_cos, const float* rope_sin, int32_t rot_dim,
                               const float* bias, void* stream);

/* ---- small dense layers (time / class embedding path: unet_video.py:472-491,
 *      resnet.py:272-273).  y[m][n] = post( sum_k pre(x[m][k]) * w[n][k] + b[n] ), m <= 16.
 *      x,y fp32; w fp16 [n][k] (nn.Linear layout); pre/post: 0 none, 1 SiLU. */
int uav_linear_small(const float* x, const void* w, const float* b, float* y,
                     int32_t m, int32_t k, int32_t n, int32_t pre_act, int32_t post_act,
                     void* stream);
/* Timesteps(num_channels, flip_sin_to_cos, downscale_freq_shift) sinusoid (diffusers 0.16
 * embeddings.get_timestep_embedding; call site unet_video.py:173,472): out fp32 [m][dim]. */
int uav_timestep_embedding(const float* t, int32_t m, int32_t dim, int32_t flip_sin_to_cos,
                           float freq_shift, float* out, void* stream);

/* ---- layout edges -------------------------------------------------------------------
 * (B,C,T,H,W) fp16/fp32 planes -> channels-last rows padded to c_pad channels, concatenating
 * up to two sources on C (unet_video.py:440), and back. */
int uav_pack_nhwc(const void* src1, int32_t c1, const void* src2, int32_t c2, int32_t src_is_f32,
                  void* dst, int32_t c_pad, int32_t n_batch, int32_t t_len, int64_t hw,
                  float scale, void* stream);
int uav_unpack_ncthw(const void* src, int32_t src_stride, int32_t src_is_f32, void* dst,
                     int32_t dst_is_f32, int32_t c, int32_t n_batch, int32_t t_len, int64_t hw,
                     float clamp_lo, float clamp_hi, void* stream);

/* ---- K9: fused classifier-free guidance + DDIM step_v0 / step_vt ---------------------
 * Replaces pipeline_upscale_a_video.py:643-645 + scheduling_ddim.py:383-433 (step_v0) and
 * :436-520 (step_vt, eta = 0).  eps: model output rows [2*n][..] (uncond | text) or [n].
 * All tensors fp16, n elements.  Coefficients are host scalars (no device->host sync).
 *   v0 : g = e_u + guidance*(e_c - e_u) ; x0 = a*sample + b*g   (+clamp)   -> writes g, x0
 *   vt : prev = c0*x0 + c1*(d0*g + d1*sample)
 * prediction types: 0 epsilon, 1 sample, 2 v_prediction (coefficients prepared by the host). */
int uav_cfg_ddim_v0(const void* eps_uncond, const void* eps_text, const void* sample,
                    void* guided_out, void* x0_out, int64_t n, float guidance,
                    float coef_sample, float coef_eps, int32_t clip, float clip_range,
                    void* stream);
int uav_ddim_vt(const void* x0, const void* guided, const void* sample, void* prev_out,
                int64_t n, float coef_x0, float coef_dir, float eps_from_model,
                float eps_from_sample, float eps_from_x0, int32_t clip, float clip_range,
                void* stream);
/* same two steps on fp32 latents / model outputs (UNetVideoModel.stream_dtype = float32: latents, the UNet output and the
 * guided output stay fp32 between DDIM steps; nothing is rounded to fp16) */
int uav_cfg_ddim_v0_f32(const float* eps_uncond, const float* eps_text, const float* sample,
                        float* guided_out, float* x0_out, int64_t n, float guidance,
                        float coef_sample, float coef_eps, int32_t clip, float clip_range,
                        void* stream);
int uav_ddim_vt_f32(const float* x0, const float* guided, const float* sample, float* prev_out,
                    int64_t n, float coef_x0, float coef_dir, float eps_from_model,
                    float eps_from_sample, float eps_from_x0, int32_t clip, float clip_range,
                    void* stream);
/* y = a*x + b*z  (add_noise: scheduling_ddim.py:524-545; window blend pipeline:630-634) */
int uav_axpby_f16(const void* x, const void* z, void* y, int64_t n, float a, float b,
                  void* stream);

/* fp32 rows -> fp16 rows (fp32 residual stream of the VAE decoder feeding a conv operand; x, y 16-B aligned) */
int uav_cast_f32_f16(const float* x, void* y, int64_t n, void* stream);
/* ABI v5.  fp32 rows [rows][c] -> fp16 rows [rows][2c] = [hi | lo], hi = fp16(x), lo = fp16(x - hi): the fp32 residual stream as
 * two MFMA operands (K doubled, weights repeated along C_in) for the convs that read it directly — the down / up samplers
 * (resnet.py:104-197) with UAV_SAMPLER_HILO=1; c % 8 == 0. */
int uav_cast_f32_hilo(const float* x, void* y, int64_t rows, int32_t c, void* stream);
/* SFT fusion of the video VAE (Fuse_sft_block, resnet.py:76-78): out = dec + w*(dec*scale + shift), elementwise;
 * in_f32 / out_f32 select fp32 instead of fp16 for the three inputs / the output */
int uav_sft_fuse(const void* dec, const void* scale, const void* shift, void* out, int64_t n, float w,
                 int32_t in_f32, int32_t out_f32, void* stream);

/* ---- K12: colour correction of the decoded frames (CLI --color_fix AdaIn | Wavelet) --------------------
 * fp32 planes: a (T,C,H,W) tensor is `planes` = T*C images of h*w.  Replaces (reference
 * models_video/color_correction.py and inference_upscale_a_video.py:322-333):
 *   F.interpolate(scale_factor=4, mode='bicubic')                 uav_resize_bicubic_f32 (ATen cubic convolution, A = -0.75,
 *                                                                 src = scale*(dst+0.5)-0.5, clamped taps; scale = in/out or 1/scale_factor)
 *   calc_mean_std (:43-57): per-plane mean and UNBIASED variance  uav_plane_stats_f32 (deterministic two-stage, fp64 combine)
 *   adaptive_instance_normalization (:59-71)                      uav_adain_apply_f32: (x-cm)/sqrt(cv+eps)*sqrt(sv+eps)+sm
 *   wavelet_blur (:73-91) + decomposition step (:101-104)         uav_atrous_blur_f32: 3x3 [1 2 1]x[1 2 1]/16, dilation `radius`,
 *                                                                 replicate padding; high_inout (optional) += src - low
 */
int64_t uav_plane_stats_workspace_bytes(int32_t planes);
int uav_plane_stats_f32(const float* x, int32_t planes, int64_t hw, float* mean_out, float* var_out,
                        void* workspace, int64_t workspace_bytes, void* stream);
int uav_adain_apply_f32(const float* x, float* out, int32_t planes, int64_t hw, const float* c_mean, const float* c_var,
                        const float* s_mean, const float* s_var, float eps, void* stream);
int uav_atrous_blur_f32(const float* src, float* low_out, float* high_inout, int32_t planes, int32_t h, int32_t w,
                        int32_t radius, void* stream);
int uav_resize_bicubic_f32(const float* src, float* dst, int32_t planes, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                           float scale_h, float scale_w, void* stream);
/* F.interpolate(mode='area') (adaptive average pooling) of fp32 planes, result multiplied by `mul`: the flow resize of
 * Propagation.forward (propagation_module.py:206-209) when flows and latents differ in resolution */
int uav_resize_area_f32(const float* src, float* dst, int32_t planes, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                        float mul, void* stream);

/* ---- K10: flow-guided propagation step -----------------------------------------------
 * Replaces one recurrence step of Propagation.forward (propagation_module.py:234-254) with
 * fbConsistencyCheck (:140-149) and flow_warp (:104-135): planar fp16 features, one frame of a
 * (C,T,H,W) tensor addressed in place through the channel strides.
 *   mask = |f + warp_bilinear(check, f)|^2 < a1*(|f|^2 + |warp(check)|^2) + a2
 *   out  = mask ? fuse*warp_{nearest|bilinear}(prev, f) + (1-fuse)*cur : cur
 * Grid arithmetic is replayed in fp16 exactly as the reference does (SURVEY.md §7 hard part 4). */
int uav_propagate_step_f16(const void* feat_prev, const void* feat_cur, const void* flow_prop,
                           const void* flow_check, void* out, int32_t c, int32_t h, int32_t w,
                           int64_t feat_chan_stride, /* elements between channel planes of prev/cur/out */
                           int64_t flow_chan_stride, /* elements between the x and y flow planes */
                           int32_t nearest,
                           int32_t coord_f16, /* 1: replay fp16 grid arithmetic; 0: fp32 */
                           float fuse_scale, float alpha1, float alpha2, void* stream);
/* ABI v5.  The same step on fp32 planes, fp32 flows and fp32 grid arithmetic — what the reference computes when its latents are
 * fp32 (pipeline_upscale_a_video.py:651 casts the flows `.to(latents)`; flow_warp builds the grid `.type_as(x)`,
 * propagation_module.py:123).  Used by the pipeline when the UNet runs on an fp32 residual stream. */
int uav_propagate_step_f32(const float* feat_prev, const float* feat_cur, const float* flow_prop,
                           const float* flow_check, float* out, int32_t c, int32_t h, int32_t w,
                           int64_t feat_chan_stride, int64_t flow_chan_stride, int32_t nearest,
                           float fuse_scale, float alpha1, float alpha2, void* stream);


/* ---- K11: RAFT optical flow (fp32; reference models_video/RAFT/*.py, raft_bi.py:26 runs it in fp32) ----
 * uav_conv_gemm_f32: same parameter struct as the fp16 kernel with 4-byte elements (a1/a2/w/residual/
 * out are fp32; c1,c2 multiples of 32 or c1==4 "small" mode; k = tap*cin_p + c; no upsample / rowbias /
 * GEGLU) on the exact-fp32 MFMA v_mfma_f32_32x32x2_f32.  Replaces nn.Conv2d of extractor.py:10-55,
 * 118-193 and update.py:6-139, and the all-pairs correlation matmul corr.py:53-60 (weights = the
 * second feature map).  Activation flags apply after bias/residual/scale. */
#define UAV_CONV_RELU     4u
#define UAV_CONV_SIGMOID  8u
#define UAV_CONV_TANH     16u
/* uav_conv_gemm_f16 only, opt-in and experimental: run the 256x256-tile kernel as one persistent workgroup per CU that
 * prefetches the first K stage of its next tile ahead of the current tile's epilogue.  Bit-identical output; measured
 * neutral on MI355X in round 1, hence off by default. */
#define UAV_CONV_PERSISTENT 64u
int uav_conv_gemm_f32(const uav_conv_params* p, void* stream);
/* nn.InstanceNorm2d (no affine) [+ReLU] on rows [n_img*hw][c] (extractor.py:27-30,48-49) */
int uav_instnorm_f32(const float* x, float* y, int32_t n_img, int32_t hw, int32_t c, float eps,
                     int32_t relu, void* stream);
int uav_add_relu_f32(const float* a, const float* b, float* out, int64_t n, int32_t relu, void* stream);
int uav_axpby_f32(const float* x, const float* z, float* y, int64_t n, float a, float b, void* stream);
/* dst[r][dst_col+j] = act(src[r][src_col+j]), j < ncols (channel concats; act 0 none, 1 relu, 4 tanh) */
int uav_copy_cols_f32(const float* src, int32_t src_stride, int32_t src_col, float* dst,
                      int32_t dst_stride, int32_t dst_col, int32_t ncols, int64_t rows, int32_t act,
                      void* stream);
/* ConvGRU gate arithmetic (update.py:44-58); zr rows hold [z | r] (2c).  mode 0: out = r*h;
 * mode 1: h <- (1-z)*h + z*q in place (out must equal h_in). */
int uav_gru_gates_f32(const float* zr, const float* h_in, const float* q, float* out, int64_t rows,
                      int32_t c, int32_t mode, void* stream);
/* bilinear plane resize, align_corners=False: RAFT_bi pre-resize of the frames to multiples of 8 (raft_bi.py:53,
 * trilinear with T unchanged) and resize_flow_pytorch (raft_bi.py:11-16); row0_scale / row1_scale multiply output
 * rows 0 and 1 (the reference's `flow[:, :, 0] *= newh/oldh; flow[:, :, 1] *= neww/oldw` indexes rows) */
int uav_resize_bilinear_f32(const float* src, float* dst, int64_t planes, int32_t hi, int32_t wi, int32_t ho,
                            int32_t wo, float row0_scale, float row1_scale, void* stream);
/* 2x2 average pooling of the correlation volume [P][h][w] (corr.py:23-27) */
int uav_avgpool2_f32(const float* src, int64_t src_stride, int32_t h, int32_t w, float* dst,
                     int64_t p_count, void* stream);
/* (2r+1)^2 x 4-level bilinear correlation lookup (corr.py:29-50): out[p][l*(2r+1)^2 + a*(2r+1) + b] */
int uav_corr_lookup_f32(const float* const* levels, const int64_t* strides, const int32_t* hs,
                        const int32_t* ws, const float* coords, int32_t coord_stride, float* out,
                        int32_t out_stride, int64_t p_count, int32_t radius, void* stream);
/* convex 8x flow upsampling (raft.py:73-85): flow rows [n*h*w][flow_stride] (x,y), mask rows
 * [n*h*w][576] -> planar (n,2,8h,8w) */
int uav_convex_upsample_f32(const float* flow, int32_t flow_stride, const float* mask, float* out,
                            int32_t n, int32_t h, int32_t w, void* stream);

/* ---- K13: the CLI's frame I/O conversions on the device ---------------------------------------------
 * frames (T,C,H,W) in 0..255 (uint8 or fp32, what utils.read_frame_from_videos returns) -> clip (C,T,H,W) fp32 in [-1,1]:
 * `(vframes/255. - 0.5) * 2` + 'b t c h w -> b c t h w' (inference_upscale_a_video.py:180,186-187); the >= 1280-px area
 * down-sampling in between (:183-184) is uav_resize_area_f32 on the (C*T) planes */
int uav_frames_to_clip_f32(const void* frames_tchw, int32_t src_is_u8, float* clip_cthw, int32_t t_len, int32_t c,
                           int64_t hw, void* stream);
/* frames (T,C,H,W) fp32 in [-1,1] -> (T,H,W,C) uint8 as the CLI hands them to imageio: `(output/2 + 0.5).clamp(0,1)*255`,
 * 't c h w -> t h w c', `.astype(np.uint8)` (truncation) (inference_upscale_a_video.py:357-359); C <= 4 */
int uav_clip_to_frames_u8(const float* frames_tchw, void* frames_thwc_u8, int32_t t_len, int32_t c, int64_t hw,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UAV_HIP_H */
