// K1/K2 — implicit-GEMM convolution / linear for gfx950 (MI355X), fp16 in, fp32 accumulate.
//
// One kernel serves every dense contraction of the UNetVideoModel / AutoencoderKLVideo hot
// path (reference ops replaced: see include/uav_hip.h):
//   per-frame 3x3 / 1x1 convs (stride 1|2, nearest-2x upsample folded into the gather),
//   temporal (k,1,1) and 3x3x3 convs (frames are the image index, zero pad at clip ends),
//   nn.Linear (1x1 "conv" over token rows).
//
// GEMM view:  D^T[n][m] = sum_k W[n][k] * X[m][k],  m = output pixel, n = output channel,
// k = tap*cin + c.  The MFMA is issued "swapped" (A operand = weights, B operand = pixels) so
// that each lane ends up owning ONE pixel m and 4 consecutive channels n per register quad:
// the epilogue (bias, time-embedding row bias, residual, scale, GEGLU) is per-lane and the
// output leaves as 8-byte (fp16) / 16-byte (fp32) vector stores into the channels-last row.
//
// Tile: 128(m) x 128(n) x 64(k) per 256-thread workgroup (4 waves, each 64x64 = 2x2 MFMA
// 32x32x16 tiles, 64 fp32 accumulators/lane).  Two LDS stages of 32 KiB, filled by
// global_load_lds DMA (16 B / lane, 1 KiB / wave-instruction): no staging VGPRs, no ds_write
// pass.  The DMA writes LDS lane-linearly, so the bank-conflict swizzle is applied on the
// per-lane SOURCE address (guide rule 21): physical 16-B slot s of LDS row r holds logical
// k-slot s ^ ((r>>1)&7); ds_read_b128 fragment reads are then conflict-free for the
// {0-3,12-15,20-27}/{4-11,16-19,28-31} lane groups of that instruction.
// Zero padding (spatial / temporal borders, M tail) is a DMA from a zero page.
//
// Roofline: MFMA-bound (arithmetic intensity 4.5*C FLOP/B for 3x3).  Algorithmic FLOP per
// launch = 2*M*N*K_logical.
//
// This header: what the kernel families share — launch arguments, the gather helpers, the epilogues.  One translation unit per family
// (round 6, VERDICT r5 #12: a k-step edit recompiles its own family, not the whole library):
//   conv_gemm128.hip   conv_gemm_kernel<SMALL>           128 x 128 tile (small grids, the 4- / 8-channel layers)
//   conv_gemm256i.hip  conv_gemm256i_kernel<6, GNK>      256 x 256 tile, 8 waves, rotated k-step (nearest-2x gather, A/B reference of the bit-identity tests)
//   conv_gemm256w.hip  conv_gemm256w_kernel<GNK>         256 x 256 tile, FOUR waves (the product kernel: every big-tile launch)
//   conv_gemm_dev.hip  round-1 family, round 2-3 loop + LayerNorm-fold instances, short-K kernel, stamped four-wave instance — compiled
//                      only with -DUAV_DEV_KERNELS (tools/ab/build_dev.sh), NOT part of libuav_hip.so
//   conv_gemm.hip      parameter checks, kernel selection, the C ABI
#pragma once
#include "uav_common.h"
#include <stdlib.h>
#include <mutex>
#include <vector>
#include <stdio.h>

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;          // 16 KiB
constexpr int B_BYTES = BN * BK * 2;          // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int LM = 256, LN = 256;
constexpr int LA_BYTES = LM * BK * 2;            // 32 KiB
constexpr int LSTAGE = 2 * LA_BYTES;             // 64 KiB (X tile + W tile)
constexpr int LEPI_BYTES = 5 * 1024;             // conv_gemm256i_kernel: staged bias (1 KiB) + 4 time-embedding row blocks

struct ConvArgs {
    const char* a1; const char* a2; int c1, c2;
    const char* w; const float* bias; const float* rowbias; int rows_per_batch, rowbias_stride;
    const char* residual; int res_stride;
    char* out; int out_stride;
    int n_img, t_len, hi, wi, ho, wo, kt, kh, kw, stride, pad_t, pad_h, pad_w, upsample;
    int n, n_pad, k_pad; float out_scale; unsigned flags;
    const char* zero_page;
    long long M;
    int korder, tile_order;
    unsigned ntiles;
    float* gn_ws; int gn_groups, gn_cpg_log2; long long gn_chunks;     // fused GroupNorm statistics (see conv_gn_store)
    int gn_cpi, gn_cstride, gn_coff;                                   // chunk placement in a workspace shared by several launches (uav_conv_params.gn_chunk_*)
    int omw, omsy, omsx, omoff;                                        // strided output rows (uav_conv_params.out_map_*)
    int a2_pix;                                                        // pixels of source 2 when it is read batch-broadcast (0: off)
    int a2_ctr;                                                        // source 2 multiplies the centre tap only (uav_conv_params.a2_center_tap)
    // LayerNorm folded into the consuming projection (uav_conv_params.ln_*): a PRODUCER also writes the fp16 rounding of its
    // fp32 result rows and, per row and 128-column chunk, (sum, sum of squares); a CONSUMER turns acc = x16 . (W o gamma) into
    // rstd_m * (acc - mu_m * colsum_n) + bias'_n with the row statistics of its operand.
    char* lnp_raw; float* lnp_stat;                                    // producer outputs (nullptr: off)
    const float* lnc_stat; const float* lnc_colsum; int lnc_chunks, lnc_n; float lnc_eps;   // consumer inputs (lnc_stat nullptr: off)
    // conv_gemm256w_kernel: byte sizes of the two sources (buffer-descriptor range = the hardware's zero fill for padding) and
    // magic numbers of the three divisions that turn a GEMM row into (image, y, x, frame) — n / d = umulhi(n, mul) >> sh, n < 2^31
    unsigned x1_bytes, x2_bytes;
    unsigned dv_hw_mul, dv_hw_sh, dv_wo_mul, dv_wo_sh, dv_t_mul, dv_t_sh;
    unsigned long long* trace;                                          // development (UAV_CONV_W4_TRACE): 8 words per workgroup
};

// ---- launch functions of the kernel families (one translation unit each; grid = workgroups, gn_mode = gn_mode_of(cpg_log2) or 0) ----
int conv_launch_tile128(const ConvArgs& a, long long grid, bool small, hipStream_t s);
int conv_launch_wave8(const ConvArgs& a, long long grid256, int gn_mode, hipStream_t s);
int conv_launch_wave4(const ConvArgs& a, long long grid256, int gn_mode, bool hilo, hipStream_t s);
#ifdef UAV_DEV_KERNELS
// which: 0 round-1 kernel (dbg / persist variants), 1 round 2-3 loop (V = 1; gn_mode, lnf = 1 LayerNorm-fold producer / 2 consumer),
// 2 short-K kernel (sk_variant 1 asm / 2 compiler-scheduled), 3 stamped four-wave instance (prints the phase ticks, synchronises)
int conv_launch_dev(int which, const ConvArgs& a, long long grid, int gn_mode, int lnf, int dbg, int persist, int sk_variant, hipStream_t s);
#endif

namespace {

// Source-2 pixel of GEMM pixel px: the skip tensors of the CFG-shared UNet head exist once and serve both batch entries
// (uav_conv_params.a2_images), i.e. images a2_images .. 2*a2_images-1 read the pixels of images 0 .. a2_images-1.
UAV_DEVINL int a2_wrap(const ConvArgs& p, int px) { return (p.a2_pix && px >= p.a2_pix) ? px - p.a2_pix : px; }

// Output row of GEMM row m: m itself, or the strided placement of a sub-pixel phase (one integer division per lane and row
// block, only on the launches that ask for it).
UAV_DEVINL long long out_row(const ConvArgs& p, long long m) {
    if (!p.omw) return m;
    const int mi = (int)m, Y = mi / p.omw, x = mi - Y * p.omw;
    return (long long)Y * p.omsy + (long long)x * p.omsx + p.omoff;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

UAV_DEVINL void dma16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Shared epilogue.  After the swapped MFMA a lane owns pixel m = mw0 + mi*32 + (lane&31) and, per
// register quad g, channels n = nw0 + ni*32 + 8g + 4*(lane>>5) + j (j = 0..3): 8-byte pieces.
// Pairs of quads are exchanged between the two half-waves with v_permlane32_swap (cdna guide T21)
// so that every lane stores / loads 16 contiguous bytes: half the store instructions, 32-B
// contiguous per row per instruction.  fp32 outputs and N tails keep the 8-byte path.
UAV_DEVINL void swap_pair(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
UAV_DEVINL uint32_t pack_h2(float x, float y) {
    half2_t h = {(half_t)x, (half_t)y};
    return __builtin_bit_cast(uint32_t, h);
}
UAV_DEVINL float2_t unpack_h2(uint32_t u) {
    half2_t h = __builtin_bit_cast(half2_t, u);
    return float2_t{(float)h[0], (float)h[1]};
}

// ---------------------------------------------------------------------------------------------
// Fused GroupNorm statistics (UAV_CONV_GN_STATS): the epilogue already holds, per lane, the final fp32 values of one
// pixel row; the consumer's GroupNorm needs (sum, sum of squares) per group of `cpg` consecutive channels over all rows of
// an instance.  A wave reduces its 64-row x 128-channel tile to one (sum, sumsq) pair per group it covers and writes
// them to gn_ws[(which * groups + g) * chunks + chunk], chunk = first row / 64 — `uav_groupnorm_finalize_partials`
// (norm.hip) then reads chunk-contiguous runs.  Fixed reduction order, no atomics: deterministic.
// DPP sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63); the total is valid in lanes 16-31 / 48-63.
UAV_DEVINL float half_sum32(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));    // row_half_mirror
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));    // row_mirror
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));   // row_bcast15 -> rows 1, 3
    return v;
}
UAV_DEVINL float both_halves(float v) {        // v(lane) + v(lane ^ 32)
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    swap_pair(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// Accumulator granularity GNM (template parameter of the fast epilogues; 0 = statistics off): the lane keeps one
// (sum, sumsq) pair per column tile ni and per NG = 4 / 2 / 1 register-quad classes — GNM 1: per quad g (groups of 4 or 8
// channels), 2: per quad pair (16), 3: per column tile (32, 64, 128) — so wide groups cost 8-16 registers, not 32.
template <int GNM> struct GnAcc { static constexpr int NG = GNM == 1 ? 4 : GNM == 2 ? 2 : 1; };
__host__ __device__ inline int gn_mode_of(int cpg_log2) { return cpg_log2 <= 3 ? 1 : cpg_log2 == 4 ? 2 : 3; }

// st/sq[ni][k]: this lane's sums over its pixels (mi) of channels nw0 + ni*32 + 8g + 4*hi32 + (0..3), g in class k.
// CL = log2(channels per group), 2..7.  After the half-wave reductions every lane 16..31 of a half holds the totals;
// lane 16+i keeps value i, so ONE store instruction per statistic leaves the wave (vector memory instructions, not
// VALU, are what the epilogue is short of).
// Chunk index of the wave tile that starts at row mw0 (64 rows per chunk); remapped when several launches share one workspace.
UAV_DEVINL long long gn_chunk_index(const ConvArgs& p, long long mw0, int rows) {
    long long k = mw0 / rows;
    if (p.gn_cpi) { const int ki = (int)k, inst = ki / p.gn_cpi; k = (long long)inst * p.gn_cstride + p.gn_coff + (ki - inst * p.gn_cpi); }
    return k;
}
template <int NI, int MI, int GNM, int CL>
UAV_DEVINL void conv_gn_store_cl(const ConvArgs& p, float (&st)[NI][GnAcc<GNM>::NG], float (&sq)[NI][GnAcc<GNM>::NG],
                                 long long mw0, int nw0, int l32, int hi32) {
    constexpr int NG = GnAcc<GNM>::NG;
    constexpr int NV = CL <= 4 ? NI * NG : CL == 5 ? NI : CL == 6 ? (NI + 1) / 2 : 1;
    float a[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { a[i] = 0.f; b[i] = 0.f; }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = CL <= 4 ? ni * NG + k : CL == 5 ? ni : CL == 6 ? (ni >> 1) : 0;
            a[i] += st[ni][k]; b[i] += sq[ni][k];
        }
    float vs = 0.f, vq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = half_sum32(a[i]), q = half_sum32(b[i]);
        if (CL >= 3) { s = both_halves(s); q = both_halves(q); }
        if (l32 == 16 + i) { vs = s; vq = q; }
    }
    // CL == 2: quad 2g + hi32 of tile ni is its own group -> group (nw0 >> 2) + 2i + hi32, both halves write;
    // CL >= 3: group (nw0 >> CL) + i, the upper half writes
    const int i = l32 - 16;
    const int grp = CL == 2 ? (nw0 >> 2) + 2 * i + hi32 : (nw0 >> CL) + i;
    const bool writer = i >= 0 && i < NV && (CL == 2 || hi32 == 1) && grp < p.gn_groups;
    if (writer) {
        float* ws_s = p.gn_ws + (long long)grp * p.gn_chunks + gn_chunk_index(p, mw0, MI * 32);
        ws_s[0] = vs;
        ws_s[(long long)p.gn_groups * p.gn_chunks] = vq;
    }
}
template <int NI, int MI, int GNM>
UAV_DEVINL void conv_gn_store(const ConvArgs& p, float (&st)[NI][GnAcc<GNM>::NG], float (&sq)[NI][GnAcc<GNM>::NG],
                              long long mw0, int nw0, int l32, int hi32) {
    if constexpr (GNM == 1) {
        if (p.gn_cpg_log2 == 2) conv_gn_store_cl<NI, MI, GNM, 2>(p, st, sq, mw0, nw0, l32, hi32);
        else conv_gn_store_cl<NI, MI, GNM, 3>(p, st, sq, mw0, nw0, l32, hi32);
    } else if constexpr (GNM == 2) {
        conv_gn_store_cl<NI, MI, GNM, 4>(p, st, sq, mw0, nw0, l32, hi32);
    } else {
        // groups of 32 / 64 / 128 channels = 1 / 2 / 4 column tiles: one reduction of the per-tile sums, the wider groups
        // are sums of those (one code path; cl is wave-uniform)
        static_assert(NI == 4, "wave tile of 128 channels");
        const int cl = p.gn_cpg_log2;
        float s[NI], q[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { s[ni] = both_halves(half_sum32(st[ni][0])); q[ni] = both_halves(half_sum32(sq[ni][0])); }
        const float s01 = s[0] + s[1], s23 = s[2] + s[3], q01 = q[0] + q[1], q23 = q[2] + q[3];
        const int i = l32 - 16;
        float vs, vq;
        if (cl == 5) { vs = i == 0 ? s[0] : i == 1 ? s[1] : i == 2 ? s[2] : s[3]; vq = i == 0 ? q[0] : i == 1 ? q[1] : i == 2 ? q[2] : q[3]; }
        else if (cl == 6) { vs = i == 0 ? s01 : s23; vq = i == 0 ? q01 : q23; }
        else { vs = s01 + s23; vq = q01 + q23; }
        const int nv = 4 >> (cl - 5);
        const int grp = (nw0 >> cl) + i;
        if (i >= 0 && i < nv && hi32 == 1 && grp < p.gn_groups) {
            float* ws_s = p.gn_ws + (long long)grp * p.gn_chunks + gn_chunk_index(p, mw0, MI * 32);
            ws_s[0] = vs;
            ws_s[(long long)p.gn_groups * p.gn_chunks] = vq;
        }
    }
}

// Fast paths: the whole wave tile lies inside M and N, fp16 output, 16-B aligned rows, one time-embedding row for the
// tile.  No predicates and no flag tests inside -> ONE basic block (the generic path below has ~130 s_waitcnt and ~270
// branches; on the K = 512 linears the epilogue was 35-40 % of the kernel time, `tools/ab_conv.sh` DBG=6).  Same arithmetic
// order as the generic path: ((acc + bias) + rowbias) + residual, then * out_scale.
//
// Round 3 — vector-memory ORDER.  On gfx9 loads and stores share one in-order counter (vmcnt): a load issued after a store
// cannot be waited for before that store has been acknowledged by the L2.  The round-2 epilogue ran, per 32-column tile,
// {4 bias loads + 4 residual loads -> wait -> 4 stores}: FOUR serialized round trips per wave tile (ISA: `L x8 [vmcnt 7..0]
// S x4` four times), the residual ones to HBM.  Now
//   * ST (staged): bias and the time-embedding row of the tile come from LDS (the 256x256i kernel stages them behind its
//     two DMA stages while the first k-step's data is in flight): ds_read, i.e. lgkmcnt — no vector load at all;
//   * the residual loads are issued AHEAD of the stores: all of them at the top (fp16 residual, D = NI), or software-
//     pipelined D column tiles ahead (statistics instances / fp32 residual, whose registers do not hold everything).
// A conv without residual now ends in 16 back-to-back stores; one with a residual pays ONE round trip instead of four.
typedef __attribute__((address_space(3))) const float4_t* lds_f4ptr_t;
// (Round 5, run 24: non-temporal residual loads / result stores in the fp32 epilogues — `nt` on every global_load / store of them — cost
// 6 % of the clip, conv 5 564 -> 6 024 ms: the fp32 stream IS re-read a few launches later, from L2 / the Infinity Cache.  Plain accesses.)
UAV_DEVINL float4_t lds_f4(unsigned byte_addr) { return *(lds_f4ptr_t)(size_t)byte_addr; }

// RF32: the residual is an fp32 row (fp32 residual stream, fp16 result: a block output that is only read as an MFMA operand);
// it is loaded in the accumulators' own layout (one float4 per register quad), no half-wave exchange.
// lb / lr: LDS byte addresses of the staged bias / time-embedding row at this wave's first column (ST only).
// Row statistics of a LayerNorm-folded consumer: mean and 1/std of the operand rows this lane owns, from the producer's
// per-chunk (sum, sum of squares) partials [chunk][row][2].
template <int MI>
UAV_DEVINL void ln_row_stats(const ConvArgs& p, long long mw0, int l32, float (&mu)[MI], float (&rstd)[MI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long long m = mw0 + mi * 32 + l32;
        float s1 = 0.f, s2 = 0.f;
        for (int c = 0; c < p.lnc_chunks; ++c) {
            const float2_t v = *(const float2_t*)(p.lnc_stat + ((long long)c * p.M + m) * 2);
            s1 += v[0]; s2 += v[1];
        }
        const float inv_n = 1.0f / (float)p.lnc_n;
        const float mean = s1 * inv_n;
        float var = s2 * inv_n - mean * mean; var = var > 0.f ? var : 0.f;
        mu[mi] = mean; rstd[mi] = rsqrtf(var + p.lnc_eps);
    }
}

// LNC: LayerNorm folded in (staged kernels only): lr holds colsum(W') instead of a time-embedding row, bias = W.beta + b.
template <int NI, int MI, bool RES, bool BIAS, bool RB, int GNM, bool RF32 = false, bool ST = false, bool LNC = false>
UAV_DEVINL void conv_epilogue_fast(const ConvArgs& p, float16_t (&acc)[NI][MI], long long mw0, int nw0, int l32, int hi32,
                                   const float* rbrow, unsigned lb = 0, unsigned lr = 0) {
    constexpr bool GN = GNM != 0;
    float lmu[MI], lrs[MI];
    if (LNC) ln_row_stats<MI>(p, mw0, l32, lmu, lrs);
    constexpr int NG = GnAcc<GNM>::NG;
    constexpr bool R16 = RES && !RF32, R32 = RES && RF32;
    constexpr int D = R16 ? (GN ? 2 : NI) : 1;            // residual prefetch distance in column tiles
    float gst[GN ? NI : 1][NG], gsq[GN ? NI : 1][NG];    // GroupNorm partial sums of the values stored (fp32, before rounding)
    if (GN) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int k = 0; k < NG; ++k) { gst[ni][k] = 0.f; gsq[ni][k] = 0.f; }
    }
    char* orow[MI];
    const char* rrow[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long long m = mw0 + mi * 32 + l32;
        orow[mi] = p.out + (out_row(p, m) * p.out_stride + nw0 + 8 * hi32) * 2;
        rrow[mi] = !RES ? nullptr : RF32 ? p.residual + (m * p.res_stride + nw0 + 4 * hi32) * 4
                                         : p.residual + (m * p.res_stride + nw0 + 8 * hi32) * 2;
    }
    const float* bptr = BIAS ? p.bias + nw0 + 4 * hi32 : nullptr;
    const float* rptr = RB ? rbrow + nw0 + 4 * hi32 : nullptr;
    const float osc = p.out_scale;
    uint4_t R[R16 ? NI : 1][MI][2];
    float4_t RF[R32 ? NI : 1][MI][4];
    auto issue_res = [&](int ni) {
        if (R16) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) R[R16 ? ni : 0][mi][gp] = *(const uint4_t*)(rrow[mi] + (ni * 32 + 16 * gp) * 2);
        }
        if (R32) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) RF[R32 ? ni : 0][mi][g] = *(const float4_t*)(rrow[mi] + (ni * 32 + 8 * g) * 4);
        }
    };
    if (RES) {
#pragma unroll
        for (int ni = 0; ni < D && ni < NI; ++ni) issue_res(ni);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        // the next residual tile goes out BEFORE this tile's stores (in-order vmcnt), D tiles ahead of its use
        if (RES && ni + D < NI) issue_res(ni + D);
        float4_t bq[4], rq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = ni * 32 + 8 * g;
            if (BIAS) bq[g] = ST ? lds_f4(lb + (co + 4 * hi32) * 4) : *(const float4_t*)(bptr + co);
            if (RB || LNC) rq[g] = ST ? lds_f4(lr + (co + 4 * hi32) * 4) : *(const float4_t*)(rptr + co);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t Rr[4] = {0, 0, 0, 0};
                if (R16) {
                    const uint4_t r = R[R16 ? ni : 0][mi][gp];
                    Rr[0] = r[0]; Rr[1] = r[1]; Rr[2] = r[2]; Rr[3] = r[3];
                    swap_pair(Rr[0], Rr[2]); swap_pair(Rr[1], Rr[3]);     // -> Rr[0..1]: quad 2gp, Rr[2..3]: quad 2gp+1
                }
                uint32_t A[2], B[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int g = 2 * gp + q;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
                    if (LNC) {                  // rstd * (acc - mu * colsum), then + (W.beta + b) below
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = lrs[mi] * (v[j] - lmu[mi] * rq[g][j]);
                    }
                    if (BIAS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += bq[g][j];
                    }
                    if (RB) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += rq[g][j];
                    }
                    if (R16) {
                        float2_t r0 = unpack_h2(Rr[2 * q]), r1 = unpack_h2(Rr[2 * q + 1]);
                        v[0] += r0[0]; v[1] += r0[1]; v[2] += r1[0]; v[3] += r1[1];
                    }
                    if (R32) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += RF[R32 ? ni : 0][mi][g][j];
                    }
                    uint32_t* d = q == 0 ? A : B;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= osc;
                    d[0] = pack_h2(v[0], v[1]);
                    d[1] = pack_h2(v[2], v[3]);
                    if (GN) {
                        constexpr int sh = GNM == 1 ? 0 : GNM == 2 ? 1 : 2;
                        gst[GN ? ni : 0][g >> sh] += (v[0] + v[1]) + (v[2] + v[3]);
                        gsq[GN ? ni : 0][g >> sh] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
                swap_pair(A[0], B[0]); swap_pair(A[1], B[1]);
                uint4_t o = {A[0], A[1], B[0], B[1]};
                *(uint4_t*)(orow[mi] + (ni * 32 + 16 * gp) * 2) = o;
            }
    }
    if constexpr (GN) conv_gn_store<NI, MI, GNM>(p, gst, gsq, mw0, nw0, l32, hi32);
}

// fp32-output fast path (fp32-stream mode of the VAE decoder / UNet: conv outputs, residual stream and GroupNorm inputs stay
// fp32, only the MFMA operands are fp16).  A lane owns pixel m and, per register quad g, 4 consecutive channels: one float4
// (16-B) store per quad straight from the accumulators, one float4 load for an fp32 residual; no half-wave exchange.
// Same arithmetic order as the fp16 paths: ((acc + bias) + rowbias) + residual, then * out_scale.
// RB: one time-embedding row for the whole wave tile (conv1 of a ResNet block whose branch tensor stays fp32).
// The residual of column tile ni + 1 is requested before tile ni's stores (see the note on vmcnt order above).
// LNP: LayerNorm-fold producer: the fp16 rounding of every result row (the consumer's MFMA operand) and the row's (sum, sum of
// squares) over this wave's 128 columns go out beside the fp32 rows.
template <int NI, int MI, bool RES, int GNM, bool RB = false, bool ST = false, bool LNP = false>
UAV_DEVINL void conv_epilogue_f32_fast(const ConvArgs& p, float16_t (&acc)[NI][MI], long long mw0, int nw0, int l32, int hi32,
                                       const float* rbrow = nullptr, unsigned lb = 0, unsigned lr = 0) {
    const float osc = p.out_scale;
    float ls1[MI], ls2[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { ls1[mi] = 0.f; ls2[mi] = 0.f; }
    constexpr bool GN = GNM != 0;
    constexpr int NG = GnAcc<GNM>::NG;
    float gst[GN ? NI : 1][NG], gsq[GN ? NI : 1][NG];
    if (GN) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int k = 0; k < NG; ++k) { gst[ni][k] = 0.f; gsq[ni][k] = 0.f; }
    }
    float* orow[MI];
    const float* rrow[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long long m = mw0 + mi * 32 + l32;
        orow[mi] = (float*)p.out + out_row(p, m) * p.out_stride + nw0 + 4 * hi32;
        rrow[mi] = RES ? (const float*)p.residual + m * p.res_stride + nw0 + 4 * hi32 : nullptr;
    }
    float4_t R[RES ? 2 : 1][MI][4];
    auto issue_res = [&](int ni) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) R[ni & 1][mi][g] = *(const float4_t*)(rrow[mi] + ni * 32 + 8 * g);
    };
    if (RES) issue_res(0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        if (RES && ni + 1 < NI) issue_res(ni + 1);
        float4_t bq[4], rq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = ni * 32 + 8 * g + 4 * hi32;
            bq[g] = ST ? lds_f4(lb + co * 4) : *(const float4_t*)(p.bias + nw0 + co);
            if (RB) rq[g] = ST ? lds_f4(lr + co * 4) : *(const float4_t*)(rbrow + nw0 + co);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[ni][mi][4 * g + j] + bq[g][j];
                    if (RB) v += rq[g][j];
                    if (RES) v += R[RES ? (ni & 1) : 0][mi][g][j];
                    o[j] = v * osc;
                }
                *(float4_t*)(orow[mi] + ni * 32 + 8 * g) = o;
                if (LNP) {
                    ls1[mi] += (o[0] + o[1]) + (o[2] + o[3]);
                    ls2[mi] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                    const long long m = mw0 + mi * 32 + l32;
                    half4_t h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                    *(half4_t*)(p.lnp_raw + (m * p.out_stride + nw0 + ni * 32 + 8 * g + 4 * hi32) * 2) = h;
                }
                if (GN) {
                    constexpr int sh = GNM == 1 ? 0 : GNM == 2 ? 1 : 2;
                    gst[GN ? ni : 0][g >> sh] += (o[0] + o[1]) + (o[2] + o[3]);
                    gsq[GN ? ni : 0][g >> sh] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                }
            }
        }
    }
    if (LNP) {                                   // the two half-waves hold the two halves of each row's channel quads
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const float a = both_halves(ls1[mi]), b = both_halves(ls2[mi]);
            if (hi32 == 0) {
                const long long m = mw0 + mi * 32 + l32;
                float2_t st = {a, b};
                *(float2_t*)(p.lnp_stat + ((long long)(nw0 >> 7) * p.M + m) * 2) = st;
            }
        }
    }
    if constexpr (GN) conv_gn_store<NI, MI, GNM>(p, gst, gsq, mw0, nw0, l32, hi32);
}

// ---------------------------------------------------------------------------------------------
// Row-coalesced fp32 epilogue of the four-wave kernel (round 5, second session).  In the accumulator layout a lane owns an
// output ROW: the 64 lanes of one global_load / store_dwordx4 of conv_epilogue_f32_fast touch 32 rows x 32 B — 32 cache lines, a
// quarter of each — and the residual is fetched ONE 32-column block ahead of its use.  The phase trace of the K = 512 linears
// (profiles/r05_w4_phase_trace_small_grids_run22.log) shows what that costs: an epilogue with an fp32 residual and an fp32
// result takes 35 k cycles per 256 x 256 tile even when three quarters of the chip are idle (54 k with the whole chip in it),
// 2.3x the tile's main loop: four serialized HBM round trips per 64-row half (4.4 k cycles each on an idle chip, 6.8 k on a
// busy one) plus 2.8 k cycles of store issue per block that do not depend on the chip's load at all — the CU's vector-memory
// path takes ~3 cycles per lane-line, whatever the line's fill.
// Here the wave first DUMPS its 64 x 128 half tile into its own 32-KiB quarter of the (now idle) stage buffers — rows of 512 B,
// the 16-B quad q of row r at physical quad q ^ (r & 7): conflict-free for the b128 writes of 8 consecutive rows (lane = row)
// and for the b128 reads of 8 lanes along a row — and the accumulators are DEAD from there on: the epilogue proper runs on a
// nearly empty register file.  It reads the tile back TRANSPOSED — lane L holds row 8k + (L >> 3), columns 4 (L & 7) .. +3 of
// each 32-column block, k = 0..7 — so a wave-wide load / store is 8 rows x 128 contiguous bytes = 8 whole cache lines instead of
// 32 quarter lines, and ALL residual loads of the half tile (32 loads, 128 VGPRs) go out before the first of them is needed: one
// round trip instead of four.  Bias, time-embedding row and residual are added per element in the same order as before
// (((acc + bias) + rowbias) + residual) * out_scale: the stored values are BIT-IDENTICAL to conv_epilogue_f32_fast.  GroupNorm
// partials are sums of the same fp32 values in another order (rows first, then the quads of a group: xor butterfly over the
// lanes) — deterministic, equal up to fp32 summation order.
// No workgroup barrier inside: LDS operations of one wave execute in order and the buffer is the wave's own; the KERNEL puts one
// barrier in front of the first dump (other waves may still be reading the stage buffers).
constexpr int CO_ROW = 512;                   // bytes per dumped row (128 fp32)
constexpr int CO_BYTES = 64 * CO_ROW;         // per wave: 32 KiB
typedef __attribute__((address_space(3))) float4_t* lds_f4wptr_t;

// ORD: loop nest of the 32 writes — quad-outermost (0) or block-outermost (1).  The same 32 instructions either way; which one hipcc
// allocates without a spill differs per kernel instance (the k-loop of this kernel sits at exactly 256 VGPRs and its accumulator
// file is full: measured, the statistics instance of 16-channel groups needs 1, the others 0 — the build audit checks all of them;
// issuing the residual loads in front of the dump, which would hide their round trip under it, spills in three of the four).
template <int ORD>
UAV_DEVINL void conv_co_dump(float16_t (&acc)[4][2], unsigned lbuf, int l32, int hi32) {
    const unsigned row = lbuf + l32 * CO_ROW;
    const int sw = l32 & 7;
    auto put = [&](int ni, int mi, int g, unsigned a) {
        float4_t v = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
        // the data operand in VGPRs: left to itself hipcc feeds ds_write_b128 from the accumulator file directly and then
        // spills the accumulators' own register class (there is not one free AGPR in this kernel)
        asm volatile("" : "+v"(v));
        *(lds_f4wptr_t)(size_t)(a + mi * 32 * CO_ROW + ni * 128) = v;
    };
    if constexpr (ORD == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned a = row + (((2 * g + hi32) ^ sw) << 4);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) put(ni, mi, g, a);
        }
    } else {
        unsigned aq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) aq[g] = row + (((2 * g + hi32) ^ sw) << 4);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) put(ni, mi, g, aq[g]);
    }
    asm volatile("" ::: "memory");
}

// Wave-uniform: does the wave tile at (mw0, nw0) take an fp32-result fast path?  Mirrors the tests of conv_epilogue exactly
// (statistics instances: the host only launches them when every wave tile qualifies, conv_gn_cpg_log2).  0: no; 1: plain;
// 2: fp32 residual; 3: time-embedding row (one batch entry per wave tile, no residual).
template <int GNK>
UAV_DEVINL int conv_co_kind(const ConvArgs& p, long long mw0, int nw0) {
    if (!(p.flags & UAV_CONV_OUT_F32)) return 0;
    if constexpr (GNK != 0) {
        if (mw0 >= p.M || nw0 >= p.n) return 0;              // (conv_epilogue returns at once for such a tile)
        return p.rowbias ? 3 : p.residual ? 2 : 1;
    } else {
        const bool rf32 = p.flags & UAV_CONV_RES_F32;
        if ((p.flags & (UAV_CONV_GELU | UAV_CONV_QUICK_GELU | UAV_CONV_GEGLU)) || !p.bias || mw0 + 64 > p.M || nw0 + 128 > p.n ||
            (p.out_stride & 3) || (p.residual && (!rf32 || (p.res_stride & 3))))
            return 0;
        if (!p.rowbias) return p.residual ? 2 : 1;
        const int b0 = (int)(mw0 / p.rows_per_batch), b1 = (int)((mw0 + 63) / p.rows_per_batch);
        return (b0 == b1 && !p.residual) ? 3 : 0;
    }
}

template <bool RES, int GNM, bool RB, bool HILO = false>
UAV_DEVINL void conv_epilogue_f32_lds(const ConvArgs& p, long long mw0, int nw0, unsigned lb, unsigned lr, unsigned lbuf) {
    constexpr int NI = 4, NK = 8;
    constexpr bool GN = GNM != 0;
    const float osc = p.out_scale;
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (HILO) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));     // (see conv_w4_epilogue)
    const int tr = lane >> 3, tq = lane & 7;                 // row 8k + tr, columns 4 tq .. 4 tq + 3 of every 32-column block
    const unsigned rbase = lbuf + tr * CO_ROW + ((tq ^ tr) << 4);     // + k * 8 rows + ni * 128 B  ((8k + tr) & 7 == tr)
    float4_t R[RES ? NI : 1][NK];
    if (RES) {
        const float* rrow0 = (const float*)p.residual + (mw0 + tr) * p.res_stride + nw0 + 4 * tq;
        const long long rstep = 8ll * p.res_stride;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int k = 0; k < NK; ++k) R[RES ? ni : 0][k] = *(const float4_t*)(rrow0 + k * rstep + ni * 32);
    }
    int orow[NK];                                            // output row of this lane's k-th row (strided for a sub-pixel phase)
#pragma unroll
    for (int k = 0; k < NK; ++k) orow[k] = (int)out_row(p, mw0 + 8 * k + tr);
    float* const obase = (float*)p.out + nw0 + 4 * tq;
    float s1[GN ? NI : 1], s2[GN ? NI : 1];
    if (GN) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { s1[ni] = 0.f; s2[ni] = 0.f; }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const float4_t bq = lds_f4(lb + (ni * 32 + 4 * tq) * 4);
        float4_t rq = {0.f, 0.f, 0.f, 0.f};
        if (RB) rq = lds_f4(lr + (ni * 32 + 4 * tq) * 4);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float4_t t = lds_f4(rbase + k * 8 * CO_ROW + ni * 128);
            float4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = t[j] + bq[j];
                if (RB) v += rq[j];
                if (RES) v += R[RES ? ni : 0][k][j];
                o[j] = v * osc;
            }
            if constexpr (HILO) {          // block tails: the fp32 value leaves as the two fp16 operands of its 1x1 consumer, hi = fp16(v), lo = fp16(v - hi)
                half4_t hv, lv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float of = o[j];
                    asm volatile("" : "+v"(of));             // the ROUNDED product v * out_scale: hipcc otherwise contracts (v * osc) - hi into one fma on the
                    hv[j] = (half_t)of;                      // unrounded product and lo differs from uav_cast_f32_hilo of the stored fp32 value (run 4, round 6)
                    lv[j] = (half_t)(of - (float)hv[j]);
                }
                half_t* const oh = (half_t*)p.out + (long long)orow[k] * p.out_stride + nw0 + 4 * tq + ni * 32;
                *(half4_t*)oh = hv;
                *(half4_t*)(oh + p.n) = lv;
            } else
                *(float4_t*)(obase + (long long)orow[k] * p.out_stride + ni * 32) = o;
            if (GN) {
                s1[GN ? ni : 0] += (o[0] + o[1]) + (o[2] + o[3]);
                s2[GN ? ni : 0] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            }
        }
    }
    asm volatile("" ::: "memory");                           // (the next dump of this wave overwrites the buffer: keep the reads above it)
    if constexpr (GN) {
        // rows: lanes that share tq (xor 8, 16, 32); then the quads of a group of 2^cl channels (xor 1, 2, 4); groups wider than a
        // 32-column block are sums of blocks.  Afterwards every lane holds the totals of its class.
        const int cl = p.gn_cpg_log2;                        // GNM 1: 2 | 3, GNM 2: 4, GNM 3: 5 | 6 | 7 (wave-uniform)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float a = s1[ni], b = s2[ni];
            a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (GNM >= 2 || cl >= 3) { a += __shfl_xor(a, 1, 64); b += __shfl_xor(b, 1, 64); }
            if (GNM >= 2) { a += __shfl_xor(a, 2, 64); b += __shfl_xor(b, 2, 64); }
            if (GNM == 3) { a += __shfl_xor(a, 4, 64); b += __shfl_xor(b, 4, 64); }
            s1[ni] = a; s2[ni] = b;
        }
        float vs, vq; int grp; bool writer;
        if (GNM == 3 && cl == 7) {
            vs = (s1[0] + s1[1]) + (s1[2] + s1[3]); vq = (s2[0] + s2[1]) + (s2[2] + s2[3]);
            grp = nw0 >> 7; writer = lane == 0;
        } else if (GNM == 3 && cl == 6) {
            vs = tr == 0 ? s1[0] + s1[1] : s1[2] + s1[3]; vq = tr == 0 ? s2[0] + s2[1] : s2[2] + s2[3];
            grp = (nw0 >> 6) + tr; writer = tr < 2 && tq == 0;
        } else {                                             // lane (tr = block, tq) writes the group its quad opens
            vs = tr == 0 ? s1[0] : tr == 1 ? s1[1] : tr == 2 ? s1[2] : s1[3];
            vq = tr == 0 ? s2[0] : tr == 1 ? s2[1] : tr == 2 ? s2[2] : s2[3];
            const int qpg = 1 << (cl - 2);                   // quads per group: 1, 2, 4, 8
            grp = (nw0 + tr * 32 + 4 * tq) >> cl; writer = tr < NI && (tq & (qpg - 1)) == 0;
        }
        if (writer && grp < p.gn_groups) {
            float* ws_s = p.gn_ws + (long long)grp * p.gn_chunks + gn_chunk_index(p, mw0, 64);
            ws_s[0] = vs;
            ws_s[(long long)p.gn_groups * p.gn_chunks] = vq;
        }
    }
}

// GEGLU fast path (same preconditions; no residual / rowbias by contract): value/gate tile pairs (2b, 2b+1).
template <int NI, int MI, bool BIAS, bool ST = false, bool LNC = false>
UAV_DEVINL void conv_epilogue_geglu_fast(const ConvArgs& p, float16_t (&acc)[NI][MI], long long mw0, int nw0, int l32, int hi32,
                                         unsigned lb = 0, unsigned lr = 0) {
    const float osc = p.out_scale;
    float lmu[MI], lrs[MI];
    if (LNC) ln_row_stats<MI>(p, mw0, l32, lmu, lrs);
#pragma unroll
    for (int blk = 0; blk < NI / 2; ++blk) {
        const int nb = nw0 + blk * 64;
        float4_t bv[4], bg[4], cv[4], cg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (BIAS) {
                bv[g] = ST ? lds_f4(lb + (blk * 64 + 8 * g + 4 * hi32) * 4) : *(const float4_t*)(p.bias + nb + 8 * g + 4 * hi32);
                bg[g] = ST ? lds_f4(lb + (blk * 64 + 32 + 8 * g + 4 * hi32) * 4) : *(const float4_t*)(p.bias + nb + 32 + 8 * g + 4 * hi32);
            }
            if (LNC) {
                cv[g] = lds_f4(lr + (blk * 64 + 8 * g + 4 * hi32) * 4);
                cg[g] = lds_f4(lr + (blk * 64 + 32 + 8 * g + 4 * hi32) * 4);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const long long m = mw0 + mi * 32 + l32;
            char* orow = p.out + (m * p.out_stride + (nb >> 1) + 8 * hi32) * 2;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t A[2], B[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int g = 2 * gp + q;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float hv = acc[2 * blk][mi][4 * g + j], gv = acc[2 * blk + 1][mi][4 * g + j];
                        if (LNC) { hv = lrs[mi] * (hv - lmu[mi] * cv[g][j]); gv = lrs[mi] * (gv - lmu[mi] * cg[g][j]); }
                        if (BIAS) { hv += bv[g][j]; gv += bg[g][j]; }
                        o[j] = hv * uav_gelu_erf(gv) * osc;
                    }
                    uint32_t* d = q == 0 ? A : B;
                    d[0] = pack_h2(o[0], o[1]); d[1] = pack_h2(o[2], o[3]);
                }
                swap_pair(A[0], B[0]); swap_pair(A[1], B[1]);
                uint4_t v = {A[0], A[1], B[0], B[1]};
                *(uint4_t*)(orow + 16 * gp * 2) = v;
            }
        }
    }
}

// GNK != 0: the kernel instance that also reduces GroupNorm statistics (accumulator granularity GNK, see GnAcc).  The
// host only launches it when every wave tile inside M x N qualifies for a fast path (conv_gn_cpg_log2), so nothing else
// is instantiated there: the statistics variants stay out of the plain kernels, whose register allocation (no scratch) is
// the one measured in DESIGN.md.
// ST: bias / time-embedding row of the tile are staged in LDS at lb / lr (byte addresses at this wave's first column).
// LNF: the LayerNorm-fold instances of the kernel (1: producer, 2: consumer) — like the statistics instances they are kernels
// of their own so that their registers do not weigh on the plain kernel's allocation; the host launches them only when every
// wave tile qualifies (conv_ln_ok).
// NF32: the caller (four-wave kernel) has already taken every wave tile that qualifies for an fp32-result fast path
// (conv_w4_epilogue / conv_co_kind, the same tests): those paths are not instantiated here.
template <int NI, int MI, int GNK = 0, bool ST = false, int LNF = 0, bool NF32 = false>
UAV_DEVINL void conv_epilogue(const ConvArgs& p, float16_t (&acc)[NI][MI], long long mw0, int nw0, int l32, int hi32,
                              unsigned lb = 0, unsigned lr = 0) {
    if constexpr (LNF == 1) {
        if (mw0 >= p.M || nw0 >= p.n) return;
        if (p.residual) conv_epilogue_f32_fast<NI, MI, true, 0, false, ST, true>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
        else conv_epilogue_f32_fast<NI, MI, false, 0, false, ST, true>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
        return;
    }
    if constexpr (LNF == 2) {
        if (mw0 >= p.M || nw0 >= p.n) return;
        if (p.flags & UAV_CONV_GEGLU) conv_epilogue_geglu_fast<NI, MI, true, ST, true>(p, acc, mw0, nw0, l32, hi32, lb, lr);
        else conv_epilogue_fast<NI, MI, false, true, false, 0, false, ST, true>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
        return;
    }
    if constexpr (GNK != 0) {
        if (mw0 >= p.M || nw0 >= p.n) return;                   // wave tile outside the output: nothing to store or count
        const float* rbrow = p.rowbias ? p.rowbias + (long long)((int)(mw0 / p.rows_per_batch)) * p.rowbias_stride : nullptr;
        if constexpr (NF32) {
            if (p.flags & UAV_CONV_OUT_F32) return;     // not reached: conv_co_kind != 0 for every such tile of a statistics instance
        } else if (p.flags & UAV_CONV_OUT_F32) {
            if (rbrow) conv_epilogue_f32_fast<NI, MI, false, GNK, true, ST>(p, acc, mw0, nw0, l32, hi32, rbrow, lb, lr);   // conv1: no residual
            else if (p.residual) conv_epilogue_f32_fast<NI, MI, true, GNK, false, ST>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
            else conv_epilogue_f32_fast<NI, MI, false, GNK, false, ST>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
            return;
        }
        if (p.flags & UAV_CONV_RES_F32) {               // fp32 stream in, fp16 operand out (host: bias, no rowbias)
            conv_epilogue_fast<NI, MI, true, true, false, GNK, true, ST>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
            return;
        }
#define UAV_EPI(RES, BIAS, RB) conv_epilogue_fast<NI, MI, RES, BIAS, RB, GNK, false, ST>(p, acc, mw0, nw0, l32, hi32, rbrow, lb, lr)
        switch ((p.residual ? 4 : 0) | (p.bias ? 2 : 0) | (rbrow ? 1 : 0)) {
            case 0: UAV_EPI(false, false, false); break;
            case 1: UAV_EPI(false, false, true); break;
            case 2: UAV_EPI(false, true, false); break;
            case 3: UAV_EPI(false, true, true); break;
            case 4: UAV_EPI(true, false, false); break;
            case 5: UAV_EPI(true, false, true); break;
            case 6: UAV_EPI(true, true, false); break;
            default: UAV_EPI(true, true, true); break;
        }
#undef UAV_EPI
        return;
    }
    const bool geglu = p.flags & UAV_CONV_GEGLU;
    const bool of32 = p.flags & UAV_CONV_OUT_F32;
    const bool rf32 = p.flags & UAV_CONV_RES_F32;
    const unsigned actf = p.flags & (UAV_CONV_GELU | UAV_CONV_QUICK_GELU);      // activation: generic path only (tiny GEMMs)
    if (!NF32 && of32 && !actf && !geglu && p.bias && mw0 + MI * 32 <= p.M && nw0 + NI * 32 <= p.n && !(p.out_stride & 3) &&
        (!p.residual || (rf32 && !(p.res_stride & 3)))) {
        if (!p.rowbias) {
            if (p.residual) conv_epilogue_f32_fast<NI, MI, true, 0, false, ST>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
            else conv_epilogue_f32_fast<NI, MI, false, 0, false, ST>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
            return;
        }
        const int b0 = (int)(mw0 / p.rows_per_batch), b1 = (int)((mw0 + MI * 32 - 1) / p.rows_per_batch);
        if (b0 == b1 && !p.residual) {
            conv_epilogue_f32_fast<NI, MI, false, 0, true, ST>(p, acc, mw0, nw0, l32, hi32, p.rowbias + (long long)b0 * p.rowbias_stride, lb, lr);
            return;
        }
    }
    if (!of32 && rf32 && !actf && !geglu && p.bias && !p.rowbias && mw0 + MI * 32 <= p.M && nw0 + NI * 32 <= p.n &&
        !(p.out_stride & 7) && !(p.res_stride & 3)) {
        conv_epilogue_fast<NI, MI, true, true, false, 0, true, ST>(p, acc, mw0, nw0, l32, hi32, nullptr, lb, lr);
        return;
    }
    // wave-uniform fast-path test
    if (!of32 && !rf32 && !actf && mw0 + MI * 32 <= p.M && nw0 + NI * 32 <= p.n && !(p.out_stride & 7) &&
        (!p.residual || !(p.res_stride & 7))) {
        if (geglu) {
            if (p.bias) conv_epilogue_geglu_fast<NI, MI, true, ST>(p, acc, mw0, nw0, l32, hi32, lb);
            else conv_epilogue_geglu_fast<NI, MI, false, false>(p, acc, mw0, nw0, l32, hi32);
            return;
        }
        const float* rbrow = nullptr;
        bool uniform = true;
        if (p.rowbias) {
            const int b0 = (int)(mw0 / p.rows_per_batch), b1 = (int)((mw0 + MI * 32 - 1) / p.rows_per_batch);
            uniform = b0 == b1;
            rbrow = p.rowbias + (long long)b0 * p.rowbias_stride;
        }
        if (uniform) {
#define UAV_EPI(RES, BIAS, RB) conv_epilogue_fast<NI, MI, RES, BIAS, RB, 0, false, ST>(p, acc, mw0, nw0, l32, hi32, rbrow, lb, lr)
            const int sel = (p.residual ? 4 : 0) | (p.bias ? 2 : 0) | (rbrow ? 1 : 0);
            switch (sel) {
                case 0: UAV_EPI(false, false, false); break;
                case 1: UAV_EPI(false, false, true); break;
                case 2: UAV_EPI(false, true, false); break;
                case 3: UAV_EPI(false, true, true); break;
                case 4: UAV_EPI(true, false, false); break;
                case 5: UAV_EPI(true, false, true); break;
                case 6: UAV_EPI(true, true, false); break;
                default: UAV_EPI(true, true, true); break;
            }
#undef UAV_EPI
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long long m = mw0 + mi * 32 + l32;
        const bool mok = m < p.M;
        const long long mc = mok ? m : 0;
        const long long mo = out_row(p, mc);
        const float* rb = p.rowbias ? p.rowbias + (long long)((int)mc / p.rows_per_batch) * p.rowbias_stride : nullptr;
        if (geglu) {
            // packed rows come in blocks of [32 value | 32 gate]: tile pair (2b, 2b+1)
#pragma unroll
            for (int blk = 0; blk < NI / 2; ++blk) {
                const int nb = nw0 + blk * 64;
                const int fbase = nb >> 1;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint32_t A[2], B[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {              // q = 0: quad 2gp, q = 1: quad 2gp+1
                        const int g = 2 * gp + q;
                        const int jn = 8 * g + 4 * hi32;
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float hv = acc[2 * blk][mi][4 * g + j], gv = acc[2 * blk + 1][mi][4 * g + j];
                            if (p.bias) { hv += p.bias[nb + jn + j]; gv += p.bias[nb + 32 + jn + j]; }
                            o[j] = hv * uav_gelu_erf(gv) * p.out_scale;
                        }
                        uint32_t* d = q == 0 ? A : B;
                        d[0] = pack_h2(o[0], o[1]); d[1] = pack_h2(o[2], o[3]);
                    }
                    swap_pair(A[0], B[0]); swap_pair(A[1], B[1]);
                    const int f = fbase + 16 * gp + 8 * hi32;
                    if (mok && f < (p.n >> 1)) {
                        uint4_t v = {A[0], A[1], B[0], B[1]};
                        *(uint4_t*)(p.out + ((long long)m * p.out_stride + f) * 2) = v;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int nq = nw0 + ni * 32 + 16 * gp;               // first channel of this quad pair (wave-uniform)
                const bool wide = !of32 && !rf32 && !actf && (nq + 16 <= p.n) && !(p.out_stride & 7) && !(p.res_stride & 7);
                if (wide) {
                    const int nl = nq + 8 * hi32;                     // the 8 channels this lane loads / stores
                    uint32_t R[4] = {0, 0, 0, 0};
                    if (p.residual) {
                        if (mok) {
                            uint4_t r = *(const uint4_t*)(p.residual + ((long long)m * p.res_stride + nl) * 2);
                            R[0] = r[0]; R[1] = r[1]; R[2] = r[2]; R[3] = r[3];
                        }
                        swap_pair(R[0], R[2]); swap_pair(R[1], R[3]);   // -> R[0..1]: quad 2gp, R[2..3]: quad 2gp+1
                    }
                    uint32_t A[2], B[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int g = 2 * gp + q;
                        const int n = nw0 + ni * 32 + 8 * g + 4 * hi32;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
                        if (p.bias) {
                            float4_t b = *(const float4_t*)(p.bias + n);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += b[j];
                        }
                        if (rb) {
                            float4_t b = *(const float4_t*)(rb + n);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += b[j];
                        }
                        if (p.residual) {
                            float2_t r0 = unpack_h2(R[2 * q]), r1 = unpack_h2(R[2 * q + 1]);
                            v[0] += r0[0]; v[1] += r0[1]; v[2] += r1[0]; v[3] += r1[1];
                        }
                        uint32_t* d = q == 0 ? A : B;
                        d[0] = pack_h2(v[0] * p.out_scale, v[1] * p.out_scale);
                        d[1] = pack_h2(v[2] * p.out_scale, v[3] * p.out_scale);
                    }
                    swap_pair(A[0], B[0]); swap_pair(A[1], B[1]);
                    if (mok) {
                        uint4_t v = {A[0], A[1], B[0], B[1]};
                        *(uint4_t*)(p.out + (mo * p.out_stride + nl) * 2) = v;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int g = 2 * gp + q;
                        const int n = nw0 + ni * 32 + 8 * g + 4 * hi32;
                        if (!mok || n >= p.n) continue;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
                        if (p.bias) {
                            float4_t b = *(const float4_t*)(p.bias + n);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += b[j];
                        }
                        if (rb) {
                            float4_t b = *(const float4_t*)(rb + n);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += b[j];
                        }
                        if (actf) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                v[j] = (actf & UAV_CONV_GELU) ? uav_gelu_erf(v[j]) : v[j] / (1.0f + __expf(-1.702f * v[j]));
                        }
                        if (p.residual) {
                            if (rf32) {
                                float4_t r = *(const float4_t*)(p.residual + ((long long)m * p.res_stride + n) * 4);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += r[j];
                            } else {
                                half4_t r = *(const half4_t*)(p.residual + ((long long)m * p.res_stride + n) * 2);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= p.out_scale;
                        if (of32) {
                            float4_t o = {v[0], v[1], v[2], v[3]};
                            *(float4_t*)(p.out + (mo * p.out_stride + n) * 4) = o;
                        } else {
                            half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            *(half4_t*)(p.out + (mo * p.out_stride + n) * 2) = o;
                        }
                    }
                }
            }
        }
    }
}

// One 64 x 128 half tile of the four-wave kernel: through LDS when it takes an fp32-result fast path, else the shared epilogue.
template <int GNK, bool HILO = false>
UAV_DEVINL void conv_w4_epilogue(const ConvArgs& p, float16_t (&acc)[4][2], long long mw0, int nw0, int l32, int hi32,
                                 unsigned lb, unsigned lr, unsigned lbuf) {
    if constexpr (HILO) {
        // lane-derived address pieces of the epilogue re-derived HERE from a fresh lane id: kept live from the top of the kernel they
        // were what hipcc spilled across the k-loop in this instance (4 VGPRs; the loop itself sits at exactly 256)
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        l32 = lane_ & 31; hi32 = lane_ >> 5;
    }
    const int kind = conv_co_kind<GNK>(p, mw0, nw0);
    if (!HILO && kind == 0) { conv_epilogue<4, 2, GNK, true, 0, true>(p, acc, mw0, nw0, l32, hi32, lb, lr); return; }
    conv_co_dump<(GNK == 2) ? 1 : 0>(acc, lbuf, l32, hi32);
    if constexpr (HILO) {                           // the kernel instance of UAV_CONV_OUT_HILO launches: every wave tile has kind != 0 (conv_hilo_ok)
        conv_epilogue_f32_lds<true, 0, false, true>(p, mw0, nw0, lb, lr, lbuf);     // kind == 2 by contract: fp32 residual, no time-embedding row
        return;
    }
    if (kind == 2) conv_epilogue_f32_lds<true, GNK, false>(p, mw0, nw0, lb, lr, lbuf);
    else if (kind == 3) conv_epilogue_f32_lds<false, GNK, true>(p, mw0, nw0, lb, lr, lbuf);
    else conv_epilogue_f32_lds<false, GNK, false>(p, mw0, nw0, lb, lr, lbuf);
}

}  // namespace
