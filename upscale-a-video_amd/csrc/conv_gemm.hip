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
#include "uav_common.h"
#include <stdlib.h>
#include <mutex>
#include <vector>
#include <stdio.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;          // 16 KiB
constexpr int B_BYTES = BN * BK * 2;          // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;

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
                for (int j = 0; j < 4; ++j) { hv[j] = (half_t)o[j]; lv[j] = (half_t)(o[j] - (float)hv[j]); }
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

template <int SMALL>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi32 = lane >> 5;          // which half of the wave (k-slot parity)
    const int l32 = lane & 31;

    const unsigned n_tiles = p.n_pad / BN;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned mt = bid / n_tiles, nt = bid - mt * n_tiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    // ---- DMA role of this thread: rows r = pass*32 + (tid>>3), physical slot tid&7 ----------
    const int slot_log = (tid & 7) ^ ((tid >> 4) & 7);   // logical k-slot fetched into phys slot
    const int rbase = tid >> 3;                          // 0..31
    // per-row gather constants, branch-free validity test (same scheme as conv_gemm256_kernel)
    int rimg[4], rtl[4], rys[4], rxs[4];
    const int hw_o = p.ho * p.wo;
    const int ups = p.upsample ? 1 : 0;
    const int ylim = p.upsample ? p.ho : p.hi, xlim = p.upsample ? p.wo : p.wi;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        long long m = m0 + ps * 32 + rbase;
        const bool ok = m < p.M;
        int mm = ok ? (int)m : 0;
        int im = mm / hw_o; int rem = mm - im * hw_o;
        int yo = rem / p.wo; int xo = rem - yo * p.wo;
        rimg[ps] = im - p.pad_t; rtl[ps] = im % p.t_len - p.pad_t;
        rys[ps] = ok ? yo * p.stride - p.pad_h : -(1 << 28); rxs[ps] = xo * p.stride - p.pad_w;
    }
    const int cin = p.c1 + p.c2;
    const int khw = p.kh * p.kw;
    const int ntaps = p.kt * khw;
    const int nk = p.k_pad / BK;
    const char* wrow = p.w + ((long long)(n0 + rbase) * p.k_pad + slot_log * 8) * 2;

    // k-steps visit the K axis TAP-INNERMOST: (chunk 0: tap 0..ntaps-1), (chunk 1: ...).  The taps of a 3x3 conv
    // re-read almost the same source pixels, so consecutive k-steps of a workgroup (and of its neighbours on the
    // XCD) hit the lines the previous step just pulled into the 4 MiB L2; with the channel-innermost order the reuse
    // distance was cin/64 k-steps x 32 workgroups = 8 MB per XCD and 65 % of the X requests missed L2 (PMC run 21).
    int kdt = 0, kdy = 0, kdx = 0, ktap = 0, kc = 0;     // wave-uniform: tap and channel offset of the NEXT k-step
    int pix[4] = {-1, -1, -1, -1};
    bool pix_valid = false;

#define ISSUE128(STAGE, KS)                                                                                  \
    {                                                                                                        \
        char* sA = smem + (STAGE) * STAGE_BYTES;                                                             \
        char* sB = sA + A_BYTES;                                                                             \
        long long wk = (long long)(KS) * BK;                                                                 \
        if (SMALL) {                                                                                         \
            /* cin_p == 8: every 16-B slot is one tap of one pixel */                                        \
            const int tap = (KS) * 8 + slot_log;                                                             \
            const int dt = tap / khw; const int rem = tap - dt * khw; const int dy = rem / p.kw; const int dx = rem - dy * p.kw; \
            _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                               \
                const int tt = rtl[ps] + dt, yv = rys[ps] + dy, xv = rxs[ps] + dx;                           \
                const bool ok = (tap < ntaps) & ((unsigned)tt < (unsigned)p.t_len) & ((unsigned)yv < (unsigned)ylim) & \
                                ((unsigned)xv < (unsigned)xlim);                                             \
                const int px = ((rimg[ps] + dt) * p.hi + (yv >> ups)) * p.wi + (xv >> ups);                  \
                const char* g = ok ? p.a1 + (long long)px * 16 : p.zero_page;                                \
                dma16(g, sA + (ps * 256 + wave * 64) * 16);                                                  \
            }                                                                                                \
        } else {                                                                                             \
            if (ntaps > 1 || !pix_valid) {                                                                   \
                _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                           \
                    const int tt = rtl[ps] + kdt, yv = rys[ps] + kdy, xv = rxs[ps] + kdx;                    \
                    const bool ok = ((unsigned)tt < (unsigned)p.t_len) & ((unsigned)yv < (unsigned)ylim) &   \
                                    ((unsigned)xv < (unsigned)xlim);                                         \
                    const int px = ((rimg[ps] + kdt) * p.hi + (yv >> ups)) * p.wi + (xv >> ups);             \
                    pix[ps] = ok ? px : -1;                                                                  \
                }                                                                                            \
                pix_valid = true;                                                                            \
            }                                                                                                \
            const bool first = kc < p.c1;                                                                    \
            const char* src = first ? p.a1 : p.a2;                                                           \
            const int cs = first ? p.c1 : p.c2;                                                              \
            const int coff = (first ? kc : kc - p.c1) + slot_log * 8;                                        \
            _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                               \
                const int pxs = first ? pix[ps] : a2_wrap(p, pix[ps]);                                       \
                const char* g = pix[ps] >= 0 ? src + ((long long)pxs * cs + coff) * 2 : p.zero_page;         \
                dma16(g, sA + (ps * 256 + wave * 64) * 16);                                                  \
            }                                                                                                \
            wk = (long long)ktap * cin + kc;                                                                 \
            if (p.korder) {                                                                                  \
                ++ktap;                                                                                      \
                if (++kdx == p.kw) { kdx = 0; if (++kdy == p.kh) { kdy = 0; ++kdt; } }                       \
                if (ktap == ntaps) { ktap = 0; kdt = 0; kdy = 0; kdx = 0; kc += BK; }                        \
            } else {                                                                                         \
                kc += BK;                                                                                    \
                if (kc >= cin) { kc = 0; ++ktap; if (++kdx == p.kw) { kdx = 0; if (++kdy == p.kh) { kdy = 0; ++kdt; } } } \
            }                                                                                                \
        }                                                                                                    \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                                     \
            dma16(wrow + ((long long)ps * 32 * p.k_pad + wk) * 2, sB + (ps * 256 + wave * 64) * 16);         \
    }

    // ---- accumulators: acc[ni][mi], wave tile = rows n [wn*64,+64) x cols m [wm*64,+64) -----
    const int wn = wave & 1, wm = wave >> 1;
    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes inside a stage): row*128 + ((slot ^ ((row>>1)&7))*16)
    int offW[2], offX[2], swz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int rw = wn * 64 + i * 32 + l32;
        int rx = wm * 64 + i * 32 + l32;
        offW[i] = A_BYTES + rw * 128; offX[i] = rx * 128;
        swz[i] = 0;
    }
    const int swW0 = ((wn * 64 + l32) >> 1) & 7, swW1 = ((wn * 64 + 32 + l32) >> 1) & 7;
    const int swX0 = ((wm * 64 + l32) >> 1) & 7, swX1 = ((wm * 64 + 32 + l32) >> 1) & 7;
    (void)swz;

    ISSUE128(0, 0)
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ks + 1 < nk) ISSUE128(cur ^ 1, ks + 1)
        const char* st = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int slot = kk * 2 + hi32;
            half8_t w0 = *(const half8_t*)(st + offW[0] + ((slot ^ swW0) << 4));
            half8_t w1 = *(const half8_t*)(st + offW[1] + ((slot ^ swW1) << 4));
            half8_t x0 = *(const half8_t*)(st + offX[0] + ((slot ^ swX0) << 4));
            half8_t x1 = *(const half8_t*)(st + offX[1] + ((slot ^ swX1) << 4));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, acc[1][1], 0, 0, 0);
        }
        cur ^= 1;
    }

#undef ISSUE128
    conv_epilogue<2, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, l32, hi32);
}

// ---------------------------------------------------------------------------------------------
// Large-tile variant: 256(m) x 256(n) x 64(k) per 512-thread workgroup (8 waves; wave tile
// 128(n) x 64(m) = 4x2 MFMA 32x32x16 tiles, 128 fp32 accumulators/lane), two 64-KiB LDS stages,
// one workgroup per CU.  Compared with the 128x128 kernel each wave issues 2x the MFMAs per
// global_load_lds instruction (4:1) and 0.75 ds_read_b128 per MFMA instead of 1, and there are
// 2x the MFMAs between two barriers.  Because only 2 waves share a SIMD, latency is hidden INSIDE
// the wave: fragments are double-buffered in registers (the reads of k-slice kk+1 are issued
// before the MFMAs of slice kk) and the DMA of the next stage is issued in the first two slices.
constexpr int LM = 256, LN = 256;
constexpr int LA_BYTES = LM * BK * 2;            // 32 KiB
constexpr int LSTAGE = 2 * LA_BYTES;             // 64 KiB (X tile + W tile)
constexpr int LEPI_BYTES = 5 * 1024;             // conv_gemm256i_kernel: staged bias (1 KiB) + 4 time-embedding row blocks

// PERSIST: the workgroup walks tiles wg, wg + gridDim.x, ... and issues the first DMA stage of its NEXT tile before
// the epilogue of the current one (both LDS stages are idle then), so the first-stage round trip hides behind it.
template <int DBG, int PERSIST = 0>   // DBG: ablation builds for profiling only (bit0: no DMA in the loop, bit1: no MFMA, 4: compiler-scheduled k-step); 0 in production
__global__ __launch_bounds__(512, 2) void conv_gemm256_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi32 = lane >> 5, l32 = lane & 31;

    const unsigned n_tiles = p.n_pad / LN;
    const int slot_log = (tid & 7) ^ ((tid >> 4) & 7);
    const int rbase = tid >> 3;                          // 0..63; rows r = pass*64 + rbase
    const int hw_o = p.ho * p.wo;
    const int ups = p.upsample ? 1 : 0;
    const int ylim = p.upsample ? p.ho : p.hi, xlim = p.upsample ? p.wo : p.wi;
    // Per-row gather constants.  Source pixel of tap (dt,dy,dx): frame rimg+dt, y = (rys+dy) >> ups, x = (rxs+dx) >> ups,
    // valid iff 0 <= rtl+dt < t_len and 0 <= rys+dy < ylim and 0 <= rxs+dx < xlim (unsigned compares); rows past M get an
    // rys that can never pass.  Everything below is branch-free: the previous formulation went through divergent
    // branches and kept its k-step counters in scratch (12 B/lane), both on the post-barrier critical path.
    int rimg[4], rtl[4], rys[4], rxs[4];
    long long m0; int n0;
    const char* wrow;
    // k-step state (wave-uniform): tap (dt,dy,dx) and channel offset of the NEXT k-step to issue
    int kdt, kdy, kdx, ktap, kc;
    int pix[4] = {-1, -1, -1, -1};
    bool pix_valid;

    // Tile id -> (m tile, n tile).  Temporal (k,1,1) / 3x3x3 convs: output frame t reads input frames t-k/2..t+k/2 at
    // the SAME pixels, so the tiles of one spatial position are made neighbours in launch order (frame index fastest):
    // the k re-reads of an input tile then come from workgroups that run together on one XCD and hit its L2.
#define SETUP_TILE(TILE)                                                                                     \
    {                                                                                                        \
        unsigned mt_ = (TILE) / n_tiles;                                                                     \
        const unsigned nt_ = (TILE) - mt_ * n_tiles;                                                         \
        if (p.kt > 1 && p.tile_order) {                                                                      \
            const unsigned hw_ = (unsigned)hw_o;                                                             \
            if (hw_ % LM == 0) {                                                                             \
                const unsigned S_ = hw_ / LM, per_clip_ = S_ * (unsigned)p.t_len;                            \
                const unsigned c_ = mt_ / per_clip_, r_ = mt_ - c_ * per_clip_;                              \
                const unsigned sp_ = r_ / (unsigned)p.t_len, t_ = r_ - sp_ * (unsigned)p.t_len;              \
                mt_ = c_ * per_clip_ + t_ * S_ + sp_;                                                        \
            }                                                                                                \
        }                                                                                                    \
        m0 = (long long)mt_ * LM;                                                                            \
        n0 = nt_ * LN;                                                                                       \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                   \
            const long long m_ = m0 + ps * 64 + rbase;                                                       \
            const bool ok_ = m_ < p.M;                                                                       \
            const int mm_ = ok_ ? (int)m_ : 0;                                                               \
            const int im_ = mm_ / hw_o; const int rem_ = mm_ - im_ * hw_o;                                   \
            const int yo_ = rem_ / p.wo; const int xo_ = rem_ - yo_ * p.wo;                                  \
            rimg[ps] = im_ - p.pad_t; rtl[ps] = im_ % p.t_len - p.pad_t;                                     \
            rys[ps] = ok_ ? yo_ * p.stride - p.pad_h : -(1 << 28); rxs[ps] = xo_ * p.stride - p.pad_w;       \
        }                                                                                                    \
        wrow = p.w + ((long long)(n0 + rbase) * p.k_pad + slot_log * 8) * 2;                                 \
        kdt = 0; kdy = 0; kdx = 0; ktap = 0; kc = 0; pix_valid = false;                                      \
    }

    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    unsigned tile = wg;
    SETUP_TILE(tile)
    const int cin = p.c1 + p.c2;
    const int khw = p.kh * p.kw;
    const int ntaps = p.kt * khw;
    const int nk = p.k_pad / BK;

#define ISSUE_STAGE(STAGE)                                                                                   \
    {                                                                                                        \
        char* sA = smem + (STAGE) * LSTAGE;                                                                  \
        if (ntaps > 1 || !pix_valid) {                                                                       \
            _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                               \
                const int tt = rtl[ps] + kdt, yv = rys[ps] + kdy, xv = rxs[ps] + kdx;                        \
                const bool ok = ((unsigned)tt < (unsigned)p.t_len) & ((unsigned)yv < (unsigned)ylim) &       \
                                ((unsigned)xv < (unsigned)xlim);                                             \
                const int px = ((rimg[ps] + kdt) * p.hi + (yv >> ups)) * p.wi + (xv >> ups);                 \
                pix[ps] = ok ? px : -1;                                                                      \
            }                                                                                                \
            pix_valid = true;                                                                                \
        }                                                                                                    \
        const bool first = kc < p.c1;                                                                        \
        const char* xsrc = first ? p.a1 : p.a2;                                                              \
        const int xcs = first ? p.c1 : p.c2;                                                                 \
        const int xcoff = (first ? kc : kc - p.c1) + slot_log * 8;                                           \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                   \
            const int pxs = first ? pix[ps] : a2_wrap(p, pix[ps]);                                           \
            const char* g = pix[ps] >= 0 ? xsrc + ((long long)pxs * xcs + xcoff) * 2 : p.zero_page;          \
            dma16(g, sA + (ps * 512 + wave * 64) * 16);                                                      \
        }                                                                                                    \
        const long long wk = (long long)ktap * cin + kc;                                                     \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                                     \
            dma16(wrow + ((long long)ps * 64 * p.k_pad + wk) * 2, sA + LA_BYTES + (ps * 512 + wave * 64) * 16); \
        ADVANCE_K()                                                                                          \
    }
#define ADVANCE_K()                                                                                          \
    {                                                                                                        \
        if (p.korder) {                          /* tap-innermost K order (see conv_gemm_kernel) */          \
            ++ktap;                                                                                          \
            if (++kdx == p.kw) { kdx = 0; if (++kdy == p.kh) { kdy = 0; ++kdt; } }                           \
            if (ktap == ntaps) { ktap = 0; kdt = 0; kdy = 0; kdx = 0; kc += BK; }                            \
        } else {                                                                                             \
            kc += BK;                                                                                        \
            if (kc >= cin) { kc = 0; ++ktap; if (++kdx == p.kw) { kdx = 0; if (++kdy == p.kh) { kdy = 0; ++kdt; } } } \
        }                                                                                                    \
    }

    const int wn = wave & 1, wm = wave >> 1;
    float16_t acc[4][2];
#define ZERO_ACC()                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                            \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    ZERO_ACC()

    // fragment addresses: row*128 + ((slot ^ sw) << 4); all 32-row tiles share sw = (l32>>1)&7
    const int sw = (l32 >> 1) & 7;
    const int offW = LA_BYTES + (wn * 128 + l32) * 128;      // + ni*4096
    const int offX = (wm * 64 + l32) * 128;                  // + mi*4096

    half8_t fw[2][4], fx[2][2];
#define LOAD_FRAGS(SET, KK)                                                                      \
    {                                                                                            \
        const int so = (((KK) * 2 + hi32) ^ sw) << 4;                                            \
        if (DBG != 5 || ks == 0) {                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) fw[SET][i] = *(const half8_t*)(st + offW + i * 4096 + so); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) fx[SET][j] = *(const half8_t*)(st + offX + j * 4096 + so); \
        }                                                                                         \
    }
#define MFMA_SET(SET)                                                                            \
    {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                      \
                if (DBG & 2) { asm volatile("" ::"v"(fw[SET][i]), "v"(fx[SET][j])); }            \
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[SET][i], fx[SET][j], acc[i][j], 0, 0, 0); \
            }                                                                                    \
    }

    ISSUE_STAGE(0)
    int cur = 0;
    if constexpr (DBG == 0 || DBG == 6) {
        // Production k-loop: the 24 ds_read_b128 + 32 MFMA of one k-step are one hand-scheduled asm block.  The
        // compiler's own waitcnt insertion put `s_waitcnt lgkmcnt(0)` in front of every MFMA group (it does not
        // count LDS reads past an LDS-DMA), which exposed the LDS latency twice per k-step; here each MFMA waits
        // for exactly the fragments it consumes (LDS returns in order), and the reads of slice kk+2 are issued
        // into the registers slice kk just released.  Read order per slice: w0 x0 x1 w1 w2 w3.
        const unsigned ldsb = (unsigned)(size_t)(lptr_t)smem;
        const unsigned bW = ldsb + (wn * 128 + l32) * 128, bX = ldsb + (wm * 64 + l32) * 128;
        unsigned so[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) so[kk] = ((kk * 2 + hi32) ^ sw) << 4;
        for (;;) {                                   // tiles of this workgroup (one iteration unless PERSIST)
        for (int ks = 0; ks < nk; ++ks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned sb = cur * LSTAGE;
            if (ks + 1 < nk) ISSUE_STAGE(cur ^ 1)
            const unsigned aw0 = bW + sb + so[0], aw1 = bW + sb + so[1], aw2 = bW + sb + so[2], aw3 = bW + sb + so[3];
            const unsigned ax0 = bX + sb + so[0], ax1 = bX + sb + so[1], ax2 = bX + sb + so[2], ax3 = bX + sb + so[3];
            half8_t w00, w01, w02, w03, x00, x01, w10, w11, w12, w13, x10, x11;
#define RD(D, A, OFF) "ds_read_b128 %[" #D "], %[" #A "] offset:" #OFF "\n"
#define RDSET(S, A, AX) RD(w##S##0, A, 32768) RD(x##S##0, AX, 0) RD(x##S##1, AX, 4096) RD(w##S##1, A, 36864) RD(w##S##2, A, 40960) RD(w##S##3, A, 45056)
#define MF(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], %[" #C "]\n"
#define WT(N) "s_waitcnt lgkmcnt(" #N ")\n"
#define MFSET(S, N0, N1, N2, N3, N4)                                                           \
    WT(N0) MF(c00, w##S##0, x##S##0) WT(N1) MF(c01, w##S##0, x##S##1)                          \
    WT(N2) MF(c10, w##S##1, x##S##0) MF(c11, w##S##1, x##S##1)                                 \
    WT(N3) MF(c20, w##S##2, x##S##0) MF(c21, w##S##2, x##S##1)                                 \
    WT(N4) MF(c30, w##S##3, x##S##0) MF(c31, w##S##3, x##S##1)
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"          // nothing of the compiler's (SMEM) may be counted below
                RDSET(0, aw0, ax0) RDSET(1, aw1, ax1)
                MFSET(0, 10, 9, 8, 7, 6)
                RDSET(0, aw2, ax2)
                MFSET(1, 10, 9, 8, 7, 6)
                RDSET(1, aw3, ax3)
                MFSET(0, 10, 9, 8, 7, 6)
                MFSET(1, 4, 3, 2, 1, 0)
                : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),
                  [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]),
                  [w00] "=&v"(w00), [w01] "=&v"(w01), [w02] "=&v"(w02), [w03] "=&v"(w03), [x00] "=&v"(x00), [x01] "=&v"(x01),
                  [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [x10] "=&v"(x10), [x11] "=&v"(x11)
                : [aw0] "v"(aw0), [aw1] "v"(aw1), [aw2] "v"(aw2), [aw3] "v"(aw3),
                  [ax0] "v"(ax0), [ax1] "v"(ax1), [ax2] "v"(ax2), [ax3] "v"(ax3)
                : "memory");
#undef RD
#undef RDSET
#undef MF
#undef WT
#undef MFSET
            cur ^= 1;
        }
        // the MFMAs issued last may still be in flight and the compiler cannot see them: cover the XDL-write ->
        // VALU-read hazard window before the epilogue touches the accumulators
        asm volatile("s_nop 15\ns_nop 15" ::: "memory");
        const long long em0 = m0;
        const int en0 = n0;
        bool has_next = false;
        if constexpr (PERSIST) {
            // Stage `cur` was last read one k-step ago and every wave has passed a barrier since: it is free.  Fill it
            // with k-step 0 of the next tile now; the epilogue below (global loads, ~800 VALU, stores) covers the flight.
            const unsigned next = tile + gridDim.x;
            has_next = next < p.ntiles;
            if (has_next) {
                tile = next;
                SETUP_TILE(tile)
                ISSUE_STAGE(cur)
            }
        }
        if constexpr (DBG == 6) {          // ablation: no epilogue (one dword per lane keeps the accumulators alive)
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
            if (sum == 12345.678f) *(float*)p.out = sum;
        } else {
            conv_epilogue<4, 2>(p, acc, em0 + wm * 64, en0 + wn * 128, l32, hi32);
        }
        if (!has_next) break;
        // The gather constants of the new tile are recomputed here instead of living through the epilogue (they cost
        // ~22 VGPRs on top of its ~230 and spilled); the opaque `tile` keeps the compiler from reusing the first copy.
        asm volatile("" : "+s"(tile));
        SETUP_TILE(tile)
        ADVANCE_K()                              // k-step 0 of this tile is already in flight
        ZERO_ACC()
        }
        return;
    } else {
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* st = smem + cur * LSTAGE;
        const bool more = ks + 1 < nk;
        // The whole DMA of the next stage (X gather + W rows) is issued FIRST: an ablation (run 15) showed the
        // loop is latency-bound — without MFMAs a k-step still takes 1.3 us (L2-hit DMA round trip), and
        // with the W half issued behind the first MFMA set only ~0.35 us of MFMA work was left to cover it.
        if (more && !(DBG & 1)) ISSUE_STAGE(cur ^ 1)
        // sched_barrier(0) pins the source order: without it the machine scheduler sinks every ds_read next to its
        // first use and waits lgkmcnt(0) in front of each MFMA group (checked in the ISA)
#define SB __builtin_amdgcn_sched_barrier(0);
        LOAD_FRAGS(0, 0)
        LOAD_FRAGS(1, 1) SB
        MFMA_SET(0) SB
        LOAD_FRAGS(0, 2) SB
        MFMA_SET(1) SB
        LOAD_FRAGS(1, 3) SB
        MFMA_SET(0) SB
        MFMA_SET(1)
#undef SB
        cur ^= 1;
    }
    }
#undef LOAD_FRAGS
#undef MFMA_SET
#undef ISSUE_STAGE

#undef SETUP_TILE
#undef ZERO_ACC
#undef ADVANCE_K
    conv_epilogue<4, 2>(p, acc, m0 + wm * 64, n0 + wn * 128, l32, hi32);     // ablation builds (DBG 1-5)
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 kernel, DMA INTERLEAVED with the MFMAs (round 2).  Same tile, LDS image, fragment reads, accumulator
// layout and epilogue as conv_gemm256_kernel<0>; what changes is WHERE the 8 global_load_lds of the next stage are
// issued.  The round-1 loop issued them back to back right after the barrier: VMEM issue is in order and the 8 waves of
// the workgroup push 64 x 1 KiB through the CU's one texture-address path at once, so every wave sat in its DMA issue
// block for ~1300 cycles per k-step while both waves of each SIMD had no MFMA in flight (ablation: 8.25 ms with, 5.82 ms
// without the DMA block).  Here
//   * the addresses of stage ks+1 are computed at the END of k-step ks-1, after the wave's last MFMA has issued: the
//     VALU work runs beside the matrix pipe's drain (and the partner wave's MFMAs) instead of on the post-barrier
//     critical path;
//   * the W operand needs no per-lane address arithmetic at all: scalar row base (s_add on SGPRs) + a constant 32-bit
//     lane offset (`global_load_lds_dwordx4 v, s[..]`);
//   * the 8 DMA instructions sit INSIDE the hand-scheduled k-step, one every few MFMAs (pattern V), so the address
//     path works while the matrix pipe does, and M0 (LDS destination) is written by s_add right before each.
// Stage hand-over is unchanged (2 stages, vmcnt(0) + barrier per k-step), so the numerics and the tile walk are
// bit-identical to the round-1 kernel (tests/test_fullsize_gpu.py compares them).
//
// V = 5 / 6 — ROTATED k-step (round 4).  The product loop (V = 1) hands a stage over at the k-step boundary: `vmcnt(0)` + barrier,
// THEN the first fragment reads of the new stage, THEN the first MFMA — every k-step starts with the matrix pipe empty for one LDS
// round trip of 8 waves x 12 reads (the waves' last MFMAs were issued before the barrier).  Here the single barrier of a k-step
// sits after its third MFMA slice: by then all fragment reads of stage k are complete (WAR: the buffer may be overwritten) and the
// DMA of stage k+1, issued a full k-step earlier, has landed (RAW, `vmcnt(0)`); behind the barrier the wave requests the FIRST
// fragments of stage k+1 and issues the DMA of stage k+2, and both fly while the fourth MFMA slice of stage k — operands already in
// registers — keeps the matrix pipe busy.  Still one barrier and one full drain per k-step, same LDS image, fragment reads,
// per-accumulator K order and epilogue: bit-identical.  V = 6 (THE DEFAULT since round 4: +1.1 % per clip, `r04_ab_conv_rotated_kstep_run8.log`):
// the 4 X pieces of stage k+2 behind the barrier, its 4 W pieces spread over the first MFMA slice of the next k-step; V = 5 (all 8
// pieces behind the barrier, two per MFMA pair) measured the same and is not instantiated.  V = 1 (the round 2-3 loop) stays for
// A/B (`UAV_CONV_DMAV=1`) and carries the LayerNorm-fold instances; V = 2 / 3 were round-2 DMA-slot placements.
template <int V, int GNK = 0, int LNF = 0>
__global__ __launch_bounds__(512, 2) void conv_gemm256i_kernel(ConvArgs p) {
    constexpr bool ROT = V == 5 || V == 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi32 = lane >> 5, l32 = lane & 31;

    const unsigned n_tiles = p.n_pad / LN;
    const int slot_log = (tid & 7) ^ ((tid >> 4) & 7);
    const int rbase = tid >> 3;                          // 0..63; rows r = pass*64 + rbase
    const int hw_o = p.ho * p.wo;
    const int ups = p.upsample ? 1 : 0;
    const int ylim = p.upsample ? p.ho : p.hi, xlim = p.upsample ? p.wo : p.wi;

    // tile id -> (m tile, n tile), frame-fastest for temporal taps (see conv_gemm256_kernel)
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    unsigned mt = tile / n_tiles;
    const unsigned nt = tile - mt * n_tiles;
    if (p.kt > 1 && p.tile_order) {
        const unsigned hw_ = (unsigned)hw_o;
        if (hw_ % LM == 0) {
            const unsigned S_ = hw_ / LM, per_clip_ = S_ * (unsigned)p.t_len;
            const unsigned c_ = mt / per_clip_, r_ = mt - c_ * per_clip_;
            const unsigned sp_ = r_ / (unsigned)p.t_len, t_ = r_ - sp_ * (unsigned)p.t_len;
            mt = c_ * per_clip_ + t_ * S_ + sp_;
        }
    }
    const long long m0 = (long long)mt * LM;
    const int n0 = nt * LN;
    int rimg[4], rtl[4], rys[4], rxs[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const long long m_ = m0 + ps * 64 + rbase;
        const bool ok_ = m_ < p.M;
        const int mm_ = ok_ ? (int)m_ : 0;
        const int im_ = mm_ / hw_o; const int rem_ = mm_ - im_ * hw_o;
        const int yo_ = rem_ / p.wo; const int xo_ = rem_ - yo_ * p.wo;
        rimg[ps] = im_ - p.pad_t; rtl[ps] = im_ % p.t_len - p.pad_t;
        rys[ps] = ok_ ? yo_ * p.stride - p.pad_h : -(1 << 28); rxs[ps] = xo_ * p.stride - p.pad_w;
    }
    const int cin = p.c1 + p.c2;
    // Temporal taps that fall outside the clip for EVERY row of the tile are skipped instead of multiplied with zeros
    // (a (3,1,1) conv on 8 frames spends 2 of its 24 tap-frames that way, a (5,1,1) conv 6 of 40, the decoder's 3x3x3
    // conv on a 3-frame chunk 2 of 9): when a frame is a whole number of m-tiles, all rows of this tile share the frame
    // t, and the valid dt are the contiguous range [dt_lo, dt_hi).  Adding the skipped zeros would not change a bit.
    int dt_lo = 0, dt_hi = p.kt;
    if (p.kt > 1 && p.korder && hw_o % LM == 0 && m0 < p.M) {
        const int t_ = (int)(m0 / hw_o) % p.t_len;
        dt_lo = p.pad_t - t_ > 0 ? p.pad_t - t_ : 0;
        dt_hi = p.t_len + p.pad_t - t_ < p.kt ? p.t_len + p.pad_t - t_ : p.kt;
    }
    const int tap_lo = dt_lo * p.kh * p.kw;
    const int ntaps = dt_hi * p.kh * p.kw;               // one past the last tap this tile multiplies
    // a2_ctr: the channel blocks of source 2 visit the centre tap only (their other weight entries are zero by contract)
    const int ctr_tap = (p.pad_t * p.kh + p.pad_h) * p.kw + p.pad_w;
    const int nk = p.a2_ctr ? (p.c1 / BK) * (ntaps - tap_lo) + p.c2 / BK
                            : (cin / BK) * (ntaps - tap_lo) + (p.k_pad - p.kt * p.kh * p.kw * cin) / BK;
    // W operand: scalar base of piece ps at K offset kb = wtile + ps*wps + kb, per-lane constant byte offset woff
    const char* wtile = p.w + (long long)n0 * p.k_pad * 2;
    const long long wps = 64ll * p.k_pad * 2;
    const unsigned woff = (unsigned)(((long long)rbase * p.k_pad + slot_log * 8) * 2);
    int kdt = dt_lo, kdy = 0, kdx = 0, ktap = tap_lo, kc = 0;     // wave-uniform: tap / channel offset of the NEXT stage to address
    const char* gx0; const char* gx1; const char* gx2; const char* gx3;
    long long wkb;

#define XADDR(PS, G)                                                                                         \
    {                                                                                                        \
        const int tt = rtl[PS] + kdt, yv = rys[PS] + kdy, xv = rxs[PS] + kdx;                                \
        const bool ok = ((unsigned)tt < (unsigned)p.t_len) & ((unsigned)yv < (unsigned)ylim) &               \
                        ((unsigned)xv < (unsigned)xlim);                                                     \
        const int px0 = ((rimg[PS] + kdt) * p.hi + (yv >> ups)) * p.wi + (xv >> ups);                        \
        const int px = first ? px0 : a2_wrap(p, px0);                                                        \
        const long long d = (xsrc - p.zero_page) + ((long long)px * xcs + xcoff) * 2;                        \
        G = p.zero_page + (ok ? d : 0ll);                                                                    \
    }
#define COMPUTE_ADDR()                                                                                       \
    {                                                                                                        \
        const bool first = kc < p.c1;                                                                        \
        const char* xsrc = first ? p.a1 : p.a2;                                                              \
        const int xcs = first ? p.c1 : p.c2;                                                                 \
        const int xcoff = (first ? kc : kc - p.c1) + slot_log * 8;                                           \
        XADDR(0, gx0) XADDR(1, gx1) XADDR(2, gx2) XADDR(3, gx3)                                              \
        wkb = ((long long)ktap * cin + kc) * 2;                                                              \
        if (p.a2_ctr && kc >= p.c1) {            /* source 2: one (centre) tap per channel block */          \
            kc += BK;                                                                                        \
        } else if (p.korder) {                   /* tap-innermost K order (see conv_gemm_kernel) */          \
            ++ktap;                                                                                          \
            if (++kdx == p.kw) { kdx = 0; if (++kdy == p.kh) { kdy = 0; ++kdt; } }                           \
            if (ktap == ntaps) {                                                                             \
                ktap = tap_lo; kdt = dt_lo; kdy = 0; kdx = 0; kc += BK;                                      \
                if (p.a2_ctr && kc >= p.c1) { ktap = ctr_tap; kdt = p.pad_t; kdy = p.pad_h; kdx = p.pad_w; } \
            }                                                                                                \
        } else {                                                                                             \
            kc += BK;                                                                                        \
            if (kc >= cin) { kc = 0; ++ktap; if (++kdx == p.kw) { kdx = 0; if (++kdy == p.kh) { kdy = 0; ++kdt; } } } \
        }                                                                                                    \
    }

    const int wn = wave & 1, wm = wave >> 1;
    float16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l32 >> 1) & 7;
    const unsigned ldsb = (unsigned)(size_t)(lptr_t)smem;
    const unsigned bW = ldsb + (wn * 128 + l32) * 128, bX = ldsb + (wm * 64 + l32) * 128;
    unsigned so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = ((kk * 2 + hi32) ^ sw) << 4;
    const unsigned ldsw = ldsb + wave * 1024;           // this wave's 1-KiB slice inside every 8-KiB piece

    // Epilogue constants -> LDS behind the two DMA stages (LEPI_BYTES): bias[n0 .. n0+256) and, per 64-row block of the
    // tile, the time-embedding row of that block's batch entry.  Requested here, written to LDS after the prologue DMA
    // has been issued (their latencies overlap) and published by the k-loop's first barrier; the fast epilogues then
    // need no vector load for them (conv_epilogue_fast: loads issued after a store wait for that store on gfx9).
    const unsigned ldsepi = ldsb + 2 * LSTAGE;
    float4_t stg = {0.f, 0.f, 0.f, 0.f};
    unsigned stg_dst = 0;                                // 0 = this thread stages nothing
    const float* stg_src = nullptr;
    if (tid < 64) {
        if (p.bias) { stg_src = p.bias + n0 + 4 * tid; stg_dst = ldsepi + tid * 16; }
    } else if (LNF == 2 && tid < 128) {              // LayerNorm-fold consumer: colsum(W') of the tile's columns takes row block 0
        const int piece = tid - 64;
        stg_src = p.lnc_colsum + n0 + 4 * piece;
        stg_dst = ldsepi + 1024 + piece * 16;
    } else if (tid < 320 && p.rowbias) {
        const int blk = (tid - 64) >> 6, piece = (tid - 64) & 63;
        long long mrow = m0 + blk * 64; if (mrow >= p.M) mrow = 0;
        const int col = n0 + 4 * piece;
        if (col + 4 <= p.n) stg_src = p.rowbias + (long long)((int)(mrow / p.rows_per_batch)) * p.rowbias_stride + col;
        stg_dst = ldsepi + 1024 + blk * 1024 + piece * 16;
    }
    if (stg_src) {
        // ROT: by inline asm — hipcc must not know this load, or it puts `s_waitcnt vmcnt(0)` in front of the LDS store below and
        // drains the two stages of DMA issued in between (the kernel counts vmcnt itself)
        if (ROT) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg) : "v"(stg_src) : "memory");
        else stg = *(const float4_t*)stg_src;
    }

    // LDS-DMA pieces of one stage: X rows ps*64.. -> +ps*8 KiB, W rows likewise behind the 32-KiB X tile.  M0 carries
    // the wave-uniform LDS destination; it is compiler-reserved, so the block saves and restores it.
#define DX(I, OFF) "s_cbranch_vccz .Lnd%=_" #I "\n" "s_add_u32 m0, %[ldsn], " #OFF "\n" "s_nop 0\n" "global_load_lds_dwordx4 %[gx" #I "], off\n" ".Lnd%=_" #I ":\n"
#define DW(I, OFF) "s_cbranch_vccz .Lnw%=_" #I "\n" "s_add_u32 m0, %[ldsn], " #OFF "\n" "s_nop 0\n" "global_load_lds_dwordx4 %[woff], %[gw" #I "]\n" ".Lnw%=_" #I ":\n"
#define D0 DX(0, 0)
#define D1 DX(1, 8192)
#define D2 DX(2, 16384)
#define D3 DX(3, 24576)
#define D4 DW(0, 32768)
#define D5 DW(1, 40960)
#define D6 DW(2, 49152)
#define D7 DW(3, 57344)
#define NO ""
#define DMA_OPERANDS                                                                                         \
    [gx0] "v"(gx0), [gx1] "v"(gx1), [gx2] "v"(gx2), [gx3] "v"(gx3), [woff] "v"(woff),                        \
    [gw0] "s"(gw0), [gw1] "s"(gw1), [gw2] "s"(gw2), [gw3] "s"(gw3), [ldsn] "s"(ldsn), [dodma] "s"(dodma)

    // ---- prologue: stage 0 -> buffer 0, addresses of stage 1 ---------------------------------
    COMPUTE_ADDR()
    {
        const char* gw0 = wtile + wkb; const char* gw1 = gw0 + wps; const char* gw2 = gw1 + wps; const char* gw3 = gw2 + wps;
        const unsigned ldsn = ldsw, dodma = __builtin_amdgcn_readfirstlane(1u);
        unsigned m0s;
        asm volatile("s_mov_b32 %[m0s], m0\n" "s_cmp_lg_u32 %[dodma], 0\n" "s_cselect_b64 vcc, -1, 0\n"
                     D0 D1 D2 D3 D4 D5 D6 D7 "s_mov_b32 m0, %[m0s]\n"
                     : [m0s] "=&s"(m0s) : DMA_OPERANDS : "memory", "scc", "vcc");
    }
    if (nk > 1) COMPUTE_ADDR()
    long long wkb_head = 0;                              // ROT, V = 6: W offset of the stage whose W pieces the next k-step's head issues
    if (ROT) {
        if (nk > 1) {                                    // stage 1 -> buffer 1 right away, then the addresses of stage 2
            const char* gw0 = wtile + wkb; const char* gw1 = gw0 + wps; const char* gw2 = gw1 + wps; const char* gw3 = gw2 + wps;
            const unsigned ldsn = ldsw + LSTAGE, dodma = __builtin_amdgcn_readfirstlane(1u);
            unsigned m0s;
            asm volatile("s_mov_b32 %[m0s], m0\n" "s_cmp_lg_u32 %[dodma], 0\n" "s_cselect_b64 vcc, -1, 0\n"
                         D0 D1 D2 D3 D4 D5 D6 D7 "s_mov_b32 m0, %[m0s]\n"
                         : [m0s] "=&s"(m0s) : DMA_OPERANDS : "memory", "scc", "vcc");
            if (nk > 2) COMPUTE_ADDR()
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // stage 0 (and the epilogue constants, older still) have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (stg_dst) *(__attribute__((address_space(3))) float4_t*)(size_t)stg_dst = stg;

#define RD(D, A, OFF) "ds_read_b128 %[" #D "], %[" #A "] offset:" #OFF "\n"
#define RDSET(S, A, AX) RD(w##S##0, A, 32768) RD(x##S##0, AX, 0) RD(x##S##1, AX, 4096) RD(w##S##1, A, 36864) RD(w##S##2, A, 40960) RD(w##S##3, A, 45056)
#define MF(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], %[" #C "]\n"
#define WT(N) "s_waitcnt lgkmcnt(" #N ")\n"
    // one 8-MFMA slice with a DMA slot after each MFMA pair
#define MFSETD(S, N0, N1, N2, N3, N4, SA, SB, SC, SD)                                          \
    WT(N0) MF(c00, w##S##0, x##S##0) WT(N1) MF(c01, w##S##0, x##S##1) SA                       \
    WT(N2) MF(c10, w##S##1, x##S##0) MF(c11, w##S##1, x##S##1) SB                              \
    WT(N3) MF(c20, w##S##2, x##S##0) MF(c21, w##S##2, x##S##1) SC                              \
    WT(N4) MF(c30, w##S##3, x##S##0) MF(c31, w##S##3, x##S##1) SD
#define KSTEP(PRE, A0, A1, A2, A3, B0, B1, B2, B3, C0, C1, C2, C3, E0, E1, E2, E3)             \
    "s_waitcnt lgkmcnt(0)\n" RDSET(0, aw0, ax0) RDSET(1, aw1, ax1) PRE                         \
    MFSETD(0, 10, 9, 8, 7, 6, A0, A1, A2, A3) RDSET(0, aw2, ax2)                               \
    MFSETD(1, 10, 9, 8, 7, 6, B0, B1, B2, B3) RDSET(1, aw3, ax3)                               \
    MFSETD(0, 10, 9, 8, 7, 6, C0, C1, C2, C3) MFSETD(1, 4, 3, 2, 1, 0, E0, E1, E2, E3)
#define ACC_OPERANDS                                                                                             \
    [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),                  \
    [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]),                  \
    [w00] "=&v"(w00), [w01] "=&v"(w01), [w02] "=&v"(w02), [w03] "=&v"(w03), [x00] "=&v"(x00), [x01] "=&v"(x01),  \
    [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [x10] "=&v"(x10), [x11] "=&v"(x11)
#define RD_OPERANDS                                                                                              \
    [aw0] "v"(aw0), [aw1] "v"(aw1), [aw2] "v"(aw2), [aw3] "v"(aw3), [ax0] "v"(ax0), [ax1] "v"(ax1), [ax2] "v"(ax2), [ax3] "v"(ax3)

    if constexpr (ROT) {
        // fragment set 0 lives ACROSS k-steps: it is requested behind the barrier of k-step k-1 (here: behind the prologue's) and
        // consumed by the first MFMA slice of k-step k
        half8_t w00, w01, w02, w03, x00, x01, w10, w11, w12, w13, x10, x11;
        __builtin_amdgcn_s_barrier();                    // stage 0 of every wave has landed, the epilogue constants are in LDS
        asm volatile("" ::: "memory");
        {
            const unsigned aw0 = bW + so[0], ax0 = bX + so[0];
            asm volatile(RDSET(0, aw0, ax0)
                         : [w00] "=&v"(w00), [w01] "=&v"(w01), [w02] "=&v"(w02), [w03] "=&v"(w03), [x00] "=&v"(x00), [x01] "=&v"(x01)
                         : [aw0] "v"(aw0), [ax0] "v"(ax0) : "memory");
        }
        int cur = 0;
        for (int ks = 0; ks < nk; ++ks) {
            const unsigned sb = cur * LSTAGE, sn = (cur ^ 1) * LSTAGE;
            const unsigned aw1 = bW + sb + so[1], aw2 = bW + sb + so[2], aw3 = bW + sb + so[3];
            const unsigned ax1 = bX + sb + so[1], ax2 = bX + sb + so[2], ax3 = bX + sb + so[3];
            const unsigned aw0 = bW + sn + so[0], ax0 = bX + sn + so[0];            // first slice of the NEXT stage (other buffer)
            // tail: DMA of stage ks+2 (X pieces; V = 5: W pieces too) into THIS k-step's buffer, released by the barrier below
            const char* gw0 = wtile + (V == 6 ? wkb_head : wkb); const char* gw1 = gw0 + wps; const char* gw2 = gw1 + wps; const char* gw3 = gw2 + wps;
            const unsigned ldsn = ldsw + cur * LSTAGE;                               // X pieces of stage ks+2 (and its W pieces, V = 5)
            const unsigned ldsh = ldsw + (cur ^ 1) * LSTAGE;                         // V = 6 head: W pieces of stage ks+1
            const unsigned dodma = __builtin_amdgcn_readfirstlane(ks + 2 < nk ? 1u : 0u);
            const unsigned dohead = __builtin_amdgcn_readfirstlane((V == 6 && ks >= 1 && ks + 1 < nk) ? 1u : 0u);
            const unsigned more = __builtin_amdgcn_readfirstlane(ks + 1 < nk ? 1u : 0u);
            unsigned m0s;
#define DWH(I, OFF) "s_cbranch_vccz .Lnh%=_" #I "\n" "s_add_u32 m0, %[ldsh], " #OFF "\n" "s_nop 0\n" "global_load_lds_dwordx4 %[woff], %[gw" #I "]\n" ".Lnh%=_" #I ":\n"
#define ROT_OPERANDS                                                                                             \
    [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),                  \
    [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]),                  \
    [w00] "+v"(w00), [w01] "+v"(w01), [w02] "+v"(w02), [w03] "+v"(w03), [x00] "+v"(x00), [x01] "+v"(x01),        \
    [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [x10] "=&v"(x10), [x11] "=&v"(x11), [m0s] "=&s"(m0s)
#define ROT_INPUTS                                                                                               \
    [aw0] "v"(aw0), [aw1] "v"(aw1), [aw2] "v"(aw2), [aw3] "v"(aw3), [ax0] "v"(ax0), [ax1] "v"(ax1), [ax2] "v"(ax2), [ax3] "v"(ax3), \
    [ldsh] "s"(ldsh), [dohead] "s"(dohead), [more] "s"(more), DMA_OPERANDS
#define ROT_BODY(H0, H1, H2, H3, T0, T1, T2, T3)                                                                 \
    "s_mov_b32 %[m0s], m0\n" "s_cmp_lg_u32 %[dohead], 0\n" "s_cselect_b64 vcc, -1, 0\n"                          \
    RDSET(1, aw1, ax1)                                                                                           \
    MFSETD(0, 10, 9, 8, 7, 6, H0, H1, H2, H3) RDSET(0, aw2, ax2)                                                 \
    MFSETD(1, 10, 9, 8, 7, 6, NO, NO, NO, NO) RDSET(1, aw3, ax3)                                                 \
    MFSETD(0, 10, 9, 8, 7, 6, NO, NO, NO, NO)                                                                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                                     \
    "s_cmp_lg_u32 %[more], 0\n" "s_cbranch_scc0 .Lnb%=\n"                                                        \
    "s_waitcnt vmcnt(0)\n" "s_barrier\n"                                                                         \
    RDSET(0, aw0, ax0)                                                                                           \
    ".Lnb%=:\n"                                                                                                  \
    "s_cmp_lg_u32 %[dodma], 0\n" "s_cselect_b64 vcc, -1, 0\n"                                                    \
    MFSETD(1, 6, 6, 6, 6, 6, T0, T1, T2, T3)                                                                     \
    "s_mov_b32 m0, %[m0s]\n"
            if constexpr (V == 5) {
                asm volatile(ROT_BODY(NO, NO, NO, NO, D0 D1, D2 D3, D4 D5, D6 D7) : ROT_OPERANDS : ROT_INPUTS : "memory", "scc", "vcc");
            } else {
                asm volatile(ROT_BODY(DWH(0, 32768), DWH(1, 40960), DWH(2, 49152), DWH(3, 57344), D0, D1, D2, D3)
                             : ROT_OPERANDS : ROT_INPUTS : "memory", "scc", "vcc");
            }
#undef DWH
#undef ROT_OPERANDS
#undef ROT_INPUTS
#undef ROT_BODY
            // addresses of stage ks+3 (its X pieces go out behind the next barrier); V = 6 keeps the W offset of stage ks+2 for the
            // next k-step's head
            wkb_head = wkb;
            if (ks + 3 < nk) COMPUTE_ADDR()
            cur ^= 1;
        }
    } else {
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned sb = cur * LSTAGE;
        const unsigned aw0 = bW + sb + so[0], aw1 = bW + sb + so[1], aw2 = bW + sb + so[2], aw3 = bW + sb + so[3];
        const unsigned ax0 = bX + sb + so[0], ax1 = bX + sb + so[1], ax2 = bX + sb + so[2], ax3 = bX + sb + so[3];
        half8_t w00, w01, w02, w03, x00, x01, w10, w11, w12, w13, x10, x11;
        // ONE asm statement for every k-step (two statements in an if/else made the register allocator shuffle the 128
        // accumulators between them: 373 spilled VGPRs); the last k-step skips its DMA slots through VCC.
        const char* gw0 = wtile + wkb; const char* gw1 = gw0 + wps; const char* gw2 = gw1 + wps; const char* gw3 = gw2 + wps;
        const unsigned ldsn = ldsw + (cur ^ 1) * LSTAGE;
        const unsigned dodma = __builtin_amdgcn_readfirstlane(ks + 1 < nk ? 1u : 0u);      // must reach the asm in an SGPR
        unsigned m0s;
#define KSTEP_STMT(...)                                                                                          \
        asm volatile("s_mov_b32 %[m0s], m0\n" "s_cmp_lg_u32 %[dodma], 0\n" "s_cselect_b64 vcc, -1, 0\n"         \
                     KSTEP(__VA_ARGS__) "s_mov_b32 m0, %[m0s]\n"                                                 \
                     : ACC_OPERANDS, [m0s] "=&s"(m0s) : RD_OPERANDS, DMA_OPERANDS : "memory", "scc", "vcc");
        if constexpr (V == 1) {            // front-loaded: 2 while the first fragments are in flight, then one per MFMA pair
            KSTEP_STMT(D0 D1, D2, D3, D4, D5, D6, D7, NO, NO, NO, NO, NO, NO, NO, NO, NO, NO)
        } else if constexpr (V == 2) {     // one DMA every 4 MFMAs over the first 28
            KSTEP_STMT(D0, NO, D1, NO, D2, NO, D3, NO, D4, NO, D5, NO, D6, NO, D7, NO, NO)
        } else {                           // V == 3: one per MFMA pair for the X gathers, then every 4 MFMAs for W
            KSTEP_STMT(D0, D1, D2, D3, NO, D4, NO, D5, NO, D6, NO, D7, NO, NO, NO, NO, NO)
        }
#undef KSTEP_STMT
        // addresses of stage ks+2: VALU beside the matrix pipe's drain, off the post-barrier critical path
        if (ks + 2 < nk) COMPUTE_ADDR()
        cur ^= 1;
    }
    }   // !ROT
#undef RD
#undef RDSET
#undef MF
#undef WT
#undef MFSETD
#undef KSTEP
#undef ACC_OPERANDS
#undef RD_OPERANDS
#undef DMA_OPERANDS
#undef DX
#undef DW
#undef D0
#undef D1
#undef D2
#undef D3
#undef D4
#undef D5
#undef D6
#undef D7
#undef NO
#undef XADDR
#undef COMPUTE_ADDR
    // the MFMAs issued last may still be in flight and the compiler cannot see them (see conv_gemm256_kernel)
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    conv_epilogue<4, 2, GNK, true, LNF>(p, acc, m0 + wm * 64, n0 + wn * 128, l32, hi32, ldsepi + wn * 512,
                                        ldsepi + 1024 + (LNF == 2 ? 0 : wm * 1024) + wn * 512);
}


// ---------------------------------------------------------------------------------------------
// conv_gemm256w_kernel (round 5): the 256 x 256 x 64 tile walked by FOUR waves, one per SIMD, each owning a 128(n) x 128(m)
// wave tile = 4 x 4 MFMA 32x32x16 tiles = 256 fp32 accumulators in the accumulator file (AGPRs), fragments, addresses and
// the epilogue in the 256 architectural VGPRs.  Why (calibration of round 5, profiles/r05_calibration_*): on this chip
// the vendor's plain fp16 GEMM of this geometry reaches 1.15-1.25 PFLOP/s on the operands the conv kernel sees where
// conv_gemm256i_kernel (8 waves, two per SIMD, 64 x 128 wave tiles) reaches 0.92-1.02; its counters show the matrix pipe
// 70 % busy at 1.64 GHz against 50-53 % at 1.8-1.9 GHz here — not the power limit: wave cycles parked at barriers /
// waitcnts (33 % vs 8 %), 1.5x the LDS fragment bytes per MFMA, twice the barrier participants, and every DMA issue /
// address instruction of one wave competing with the partner wave's MFMA issue.  This kernel keeps the LDS image, the
// swapped MFMA (lane = pixel, 4 channels per register quad), the per-accumulator K order (-> BIT-IDENTICAL results) and
// the epilogues of conv_gemm256i_kernel, and changes the schedule:
//   * one instruction stream per SIMD: the 64 MFMAs of a k-step issue back to back, everything else — 32 ds_read_b128,
//     16 LDS-DMA pieces, the gather's validity arithmetic, two barriers — sits in the issue slots between them (one asm
//     statement per k-step, self-contained: nothing asynchronous is pending in a register when it ends);
//   * 0.5 ds_read_b128 per MFMA (8 fragments feed 16 MFMAs) instead of 0.75; the whole 64-column stage lives in 128 VGPRs:
//     slices 2-3 are read during slice 0, slices 0-1 of the NEXT stage during slice 3;
//   * the gather is a buffer load: per lane and row a 32-bit byte offset computed ONCE per tile, the tap / channel-block
//     step is a scalar added to the buffer base, and padding is the hardware's out-of-range rule — a precomputed per-row
//     bit mask over the taps ORs the offset to 0xffffffff (2 VALU per piece and k-step instead of ~12, no zero page);
//   * LDS-DMA stays in flight across both barriers (counted vmcnt, never 0 while more stages follow).

struct W4Srd { unsigned w[4]; };
UAV_DEVINL uint4_t w4_srd(const char* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    uint4_t r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
UAV_DEVINL unsigned udiv_magic(unsigned n, unsigned mul, unsigned sh) { return sh >= 32u ? n : (__umulhi(n, mul) >> sh); }

// TR = 1: development instance (UAV_CONV_W4_TRACE=1) that stamps s_memtime at the phase boundaries of every workgroup.
// HILO: the instance of UAV_CONV_OUT_HILO launches (a kernel of its own: one more epilogue instantiation inside the default
// instance moved hipcc's allocation of the k-loop and spilled 4 VGPRs there — measured, round 6).
template <int GNK, int TR = 0, bool HILO = false>
__global__ __launch_bounds__(256, 1) void conv_gemm256w_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi32 = lane >> 5, l32 = lane & 31;
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (TR) ts[0] = __builtin_amdgcn_s_memtime();

    // tile id -> (m tile, n tile), frame-fastest for temporal taps (see conv_gemm256_kernel)
    const unsigned n_tiles = p.n_pad / LN;
    const int hw_o = p.ho * p.wo;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    unsigned mt = tile / n_tiles;
    const unsigned nt = tile - mt * n_tiles;
    if (p.kt > 1 && p.tile_order) {
        const unsigned hw_ = (unsigned)hw_o;
        if (hw_ % LM == 0) {
            const unsigned S_ = hw_ / LM, per_clip_ = S_ * (unsigned)p.t_len;
            const unsigned c_ = mt / per_clip_, r_ = mt - c_ * per_clip_;
            const unsigned sp_ = r_ / (unsigned)p.t_len, t_ = r_ - sp_ * (unsigned)p.t_len;
            mt = c_ * per_clip_ + t_ * S_ + sp_;
        }
    }
    const long long m0 = (long long)mt * LM;
    const int n0 = nt * LN;
    const int cin = p.c1 + p.c2;

    // ---- gather constants: byte offset of (row, slot) at tap (pad_t, pad_h, pad_w) and the mask of INVALID taps ------------
    const int slot_log = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const int rlane = wave * 8 + (lane >> 3);            // this lane's row inside every 32-row DMA piece
    // im[i]: which tap displacements fall outside the input for this lane's row of piece i — bits [0:7] frame steps dt, [8:15]
    // rows dy, [16:23] columns dx (all set for a row past M); a stage's tap selects one bit of each field (tab_sel below).  The
    // invalid steps of an axis are a prefix and a suffix of 0 .. k-1: two clamps and shifts, no loop over the taps.
    unsigned vo[8], vo2[8], im[8];
    auto axis_bad = [](int c0, int lim) -> unsigned {     // steps d in 0 .. 7 with c0 + d outside [0, lim)
        const int lo = c0 < 0 ? (-c0 < 8 ? -c0 : 8) : 0;
        const int h0 = lim - c0 < 0 ? 0 : (lim - c0 < 8 ? lim - c0 : 8);
        return ((1u << lo) - 1u) | (0xffu & ~((1u << h0) - 1u));
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long m_ = m0 + i * 32 + rlane;
        const bool ok_ = m_ < p.M;
        const unsigned mm_ = ok_ ? (unsigned)m_ : 0u;
        const unsigned img_ = udiv_magic(mm_, p.dv_hw_mul, p.dv_hw_sh);
        const unsigned rem_ = mm_ - img_ * (unsigned)hw_o;
        const unsigned yo_ = udiv_magic(rem_, p.dv_wo_mul, p.dv_wo_sh);
        const unsigned xo_ = rem_ - yo_ * (unsigned)p.wo;
        const unsigned tt_ = img_ - udiv_magic(img_, p.dv_t_mul, p.dv_t_sh) * (unsigned)p.t_len;
        const int yi_ = (int)yo_ * p.stride, xi_ = (int)xo_ * p.stride;
        const unsigned px_ = (img_ * (unsigned)p.hi + (unsigned)yi_) * (unsigned)p.wi + (unsigned)xi_;
        vo[i] = (px_ * (unsigned)p.c1 + (unsigned)slot_log * 8u) * 2u;
        const unsigned px2_ = (p.a2_pix && px_ >= (unsigned)p.a2_pix) ? px_ - (unsigned)p.a2_pix : px_;
        vo2[i] = (px2_ * (unsigned)p.c2 + (unsigned)slot_log * 8u) * 2u;
        const unsigned pk = axis_bad((int)tt_ - p.pad_t, p.t_len) | (axis_bad(yi_ - p.pad_h, p.hi) << 8) | (axis_bad(xi_ - p.pad_w, p.wi) << 16);
        im[i] = ok_ ? pk : 0x00ffffffu;
    }
    const int khw = p.kh * p.kw;
    const unsigned woff = (unsigned)(((long long)rlane * p.k_pad + slot_log * 8) * 2);
    const unsigned wps32 = (unsigned)(32ll * p.k_pad * 2);
    const char* wtile = p.w + (long long)n0 * p.k_pad * 2;

    // temporal taps outside the clip for every row of the tile are skipped (see conv_gemm256i_kernel)
    // (all of it scalar: 32-bit magic divisions — a 64-bit division would be expanded on the VALU and drag nk, and with it
    // every wave-uniform operand of the k-step below, into vector registers)
    int dt_lo = 0, dt_hi = p.kt;
    if (p.kt > 1 && hw_o % LM == 0 && m0 < p.M) {
        const unsigned img0 = udiv_magic((unsigned)m0, p.dv_hw_mul, p.dv_hw_sh);
        const int t_ = (int)(img0 - udiv_magic(img0, p.dv_t_mul, p.dv_t_sh) * (unsigned)p.t_len);
        dt_lo = p.pad_t - t_ > 0 ? p.pad_t - t_ : 0;
        dt_hi = p.t_len + p.pad_t - t_ < p.kt ? p.t_len + p.pad_t - t_ : p.kt;
    }
    dt_lo = __builtin_amdgcn_readfirstlane(dt_lo); dt_hi = __builtin_amdgcn_readfirstlane(dt_hi);
    const int tap_lo = dt_lo * khw;
    const int ntaps = dt_hi * khw;
    const int ctr_tap = (p.pad_t * p.kh + p.pad_h) * p.kw + p.pad_w;
    const int nk = __builtin_amdgcn_readfirstlane(p.a2_ctr ? (p.c1 / BK) * (ntaps - tap_lo) + p.c2 / BK : (cin / BK) * (ntaps - tap_lo));

    // ---- scalar address walk of the stages (tap-innermost K order): 32-bit, inside the asm, hidden between the MFMAs -------
    //   Both buffer descriptors are constant per source: X = (source base - xbias, source bytes + 2 xbias), W = (this tile's
    //   rows, "no limit").  A stage is addressed by two scalars: X displacement xso = xkc + tab[tap] — added to the per-lane
    //   offsets on the VALU (measured, run 12: gfx950 range-checks voffset + SOFFSET, so a displacement in the SGPR offset zero-
    //   fills valid pixels near the end of the tensor) — channel-block bytes + xbias +
    //   the tap's byte displacement (>= -xbias; pixel displacements in the lanes of `tab_pd`, lane = tap, read with v_readlane,
    //   times the source's bytes per pixel) — and W soffset
    //   = wofs (+ piece rows), stepped by one tap (cin * 2 bytes) or, behind the last tap of a channel block, back to the first
    //   tap of the next block (wwrap).  Source 2 (channel blocks >= c1; centre tap only with a2_ctr) is a second phase with its
    //   own constants, entered through a wave-uniform branch once per tile.
    const int cin2 = cin * 2;
    int tap0 = tap_lo, tapend = ntaps;                            // taps of a channel block in the current phase
    int wwrap = 128 - (ntaps - tap_lo - 1) * cin2;
    int blk = p.c1 / BK;                                          // channel blocks left in this phase
    int tau = tap_lo;
    int wofs = tap_lo * cin2;
    // tables over the taps, one tap per lane: pixel displacement (x cs2 = bytes in the current source) and the three validity
    // bits a tap selects
    int tab_sel, tab_pd;
    {
        const int tp = lane < 32 ? lane : 0;
        const int dt_ = (int)(((float)tp + 0.5f) * (1.0f / (float)khw)), r_ = tp - dt_ * khw;      // exact: tp < 32
        const int dy_ = (int)(((float)r_ + 0.5f) * (1.0f / (float)p.kw)), dx_ = r_ - dy_ * p.kw;
        tab_pd = ((dt_ - p.pad_t) * p.hi + (dy_ - p.pad_h)) * p.wi + (dx_ - p.pad_w);
        tab_sel = (1 << (dt_ & 7)) | (1 << (8 + (dy_ & 7))) | (1 << (16 + (dx_ & 7)));
    }
    const int pdmin = ((p.pad_t * p.hi + p.pad_h) * p.wi + p.pad_w);      // -(most negative pixel displacement)
    int xkc = pdmin * p.c1 * 2;                                   // channel-block bytes + xbias of the next stage to address
    int cs2 = p.c1 * 2;                                           // bytes per pixel of the current source
    uint4_t xsrd = w4_srd(p.a1 - (long long)pdmin * p.c1 * 2, p.x1_bytes + (unsigned)(2 * pdmin * p.c1 * 2));
    const uint4_t wsrd = w4_srd(wtile, 0x7fffffffu);
    // phase 2 (called when blk reaches 0, before the asm addresses the next stage)
#define W4_PHASE2()                                                                                          \
    {                                                                                                        \
        blk = 0x40000000;                                                                                    \
        if (p.c2 > 0) {                                                                                      \
            xsrd = w4_srd(p.a2 - (long long)pdmin * p.c2 * 2, p.x2_bytes + (unsigned)(2 * pdmin * p.c2 * 2)); \
            xkc = pdmin * p.c2 * 2;                                                                          \
            cs2 = p.c2 * 2;                                                                                  \
            if (p.a2_ctr) {                                                                                  \
                wofs += (ctr_tap - tap_lo) * cin2;                                                           \
                tap0 = ctr_tap; tapend = ctr_tap + 1; tau = ctr_tap; wwrap = 128;                            \
            }                                                                                                \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) vo[i] = vo2[i];                                    \
        }                                                                                                    \
    }
    // the walk itself (asm): this stage's scalars -> stap (validity bits), xso (X soffset), wso (W soffset); then advance
#define W4_WALK1                                                                                             \
    "v_readlane_b32 %[stap], %[tabsel], %[tau]\n"                                                            \
    "v_readlane_b32 %[xso], %[tab], %[tau]\n"                                                                \
    "s_mov_b32 %[wso], %[wofs]\n"                                                                            \
    "s_add_i32 %[tau], %[tau], 1\n"                                                                          \
    "s_mul_i32 %[xso], %[xso], %[cs2]\n"                                                                     \
    "s_add_u32 %[xso], %[xso], %[xkc]\n"
#define W4_WALK2                                                                                             \
    "s_cmp_eq_u32 %[tau], %[tapend]\n"                                                                       \
    "s_cselect_b32 %[tau], %[tap0], %[tau]\n"                                                                \
    "s_cselect_b32 %[sa], 128, 0\n"                                                                          \
    "s_cselect_b32 %[sb], %[wwrap], %[cin2]\n"                                                               \
    "s_cselect_b32 %[sc], -1, 0\n"
#define W4_WALK3                                                                                             \
    "s_add_u32 %[xkc], %[xkc], %[sa]\n"                                                                      \
    "s_add_u32 %[wofs], %[wofs], %[sb]\n"                                                                    \
    "s_add_i32 %[blk], %[blk], %[sc]\n"
#define W4_WALK W4_WALK1 W4_WALK2 W4_WALK3
#define W4_WALK_OUT                                                                                          \
    [tau] "+s"(tau), [xkc] "+s"(xkc), [wofs] "+s"(wofs), [blk] "+s"(blk), [stap] "=&s"(stap), [xso] "=&s"(xso),            \
    [wso] "=&s"(wso), [sa] "=&s"(sa), [sb] "=&s"(sb), [sc] "=&s"(sc)
#define W4_WALK_IN                                                                                           \
    [tab] "v"(tab_pd), [tabsel] "v"(tab_sel), [tapend] "s"(tapend), [tap0] "s"(tap0), [wwrap] "s"(wwrap), [cin2] "s"(cin2), [cs2] "s"(cs2)

    const int wn = wave & 1, wm = wave >> 1;

    const int sw = (l32 >> 1) & 7;
    const unsigned ldsb = (unsigned)(size_t)(lptr_t)smem;
    const unsigned bW = ldsb + LA_BYTES + (wn * 128 + l32) * 128, bX = ldsb + (wm * 128 + l32) * 128;
    unsigned aw0 = bW + ((((0 * 2 + hi32) ^ sw)) << 4), aw1 = bW + ((((1 * 2 + hi32) ^ sw)) << 4);
    unsigned aw2 = bW + ((((2 * 2 + hi32) ^ sw)) << 4), aw3 = bW + ((((3 * 2 + hi32) ^ sw)) << 4);
    unsigned ax0 = bX + ((((0 * 2 + hi32) ^ sw)) << 4), ax1 = bX + ((((1 * 2 + hi32) ^ sw)) << 4);
    unsigned ax2 = bX + ((((2 * 2 + hi32) ^ sw)) << 4), ax3 = bX + ((((3 * 2 + hi32) ^ sw)) << 4);
    const unsigned ldsw = ldsb + wave * 1024;            // this wave's 1-KiB slice inside every 4-KiB group of rows

    // epilogue constants as DMA pieces of their own (older than every stage piece on vmcnt): bias[n0 .. n0 + 256), then per
    // 64-row block of the tile the time-embedding row of that block's batch entry
    char* sepi = smem + 2 * LSTAGE;
    if (wave == 0 && p.bias) dma16((const char*)(p.bias + n0 + lane * 4), sepi);
    if (p.rowbias) {
        long long mrow = m0 + wave * 64; if (mrow >= p.M) mrow = 0;
        const int col = n0 + lane * 4;
        const float* r = p.rowbias + (long long)((int)(mrow / p.rows_per_batch)) * p.rowbias_stride + (col + 4 <= p.n ? col : 0);
        dma16((const char*)r, sepi + 1024 + wave * 1024);
    }

#define RD(D, A, OFF) "ds_read_b128 %[" #D "], %[" #A "] offset:" #OFF "\n"
#define MF(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], %[" #C "]\n"
    // X piece I (rows I*32 ..): the stage's tap bits against the row's invalid-step bits -> effective offset (all ones: out of
    // range, the load returns zeros), M0 = LDS destination, buffer load to LDS
#define PXA(I)                                                                                               \
    "v_and_b32 %[t" #I "], %[stap], %[im" #I "]\n"                                                           \
    "v_cmp_ne_u32_e64 %[sp], 0, %[t" #I "]\n"                                                                \
    "v_add_u32 %[t" #I "], %[xso], %[vo" #I "]\n"                                                            \
    "v_cndmask_b32_e64 %[t" #I "], %[t" #I "], -1, %[sp]\n"
#define PXB(I, OFF)                                                                                          \
    "s_cbranch_vccz .Lnx%=_" #I "\n"                                                                         \
    "s_add_u32 m0, %[ldsn], " #OFF "\n"                                                                      \
    "s_nop 0\n"                                                                                              \
    "buffer_load_dwordx4 %[t" #I "], %[xsrd], 0 offen lds\n"                                                 \
    ".Lnx%=_" #I ":\n"
#define PX(I, OFF) PXA(I) PXB(I, OFF)
    // W piece I (rows I*32 ..): every lane valid; the piece's row offset accumulates in wso
#define PW(I, OFF)                                                                                           \
    "s_cbranch_vccz .Lnw%=_" #I "\n" "s_add_u32 m0, %[ldsn], " #OFF "\n" "s_nop 0\n"                         \
    "buffer_load_dwordx4 %[woff], %[wsrd], %[wso] offen lds\n" "s_add_u32 %[wso], %[wso], %[wps]\n" ".Lnw%=_" #I ":\n"
#define W4_DMA_IN                                                                                            \
    [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]), [vo5] "v"(vo[5]),            \
    [vo6] "v"(vo[6]), [vo7] "v"(vo[7]), [im0] "v"(im[0]), [im1] "v"(im[1]), [im2] "v"(im[2]), [im3] "v"(im[3]),            \
    [im4] "v"(im[4]), [im5] "v"(im[5]), [im6] "v"(im[6]), [im7] "v"(im[7]), [woff] "v"(woff), [xsrd] "s"(xsrd),            \
    [wsrd] "s"(wsrd), [wps] "s"(wps32), [ldsn] "s"(ldsn), [dodma] "s"(dodma), W4_WALK_IN
#define W4_TMP_OUT                                                                                           \
    [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),        \
    [t7] "=&v"(t7), [m0s] "=&s"(m0s), [sp] "=&s"(spair), W4_WALK_OUT

    if (TR) ts[1] = __builtin_amdgcn_s_memtime();
    // ---- prologue: stage 0 -> buffer 0, stage 1 -> buffer 1 -----------------------------------------
    for (int st = 0; st < 2 && st < nk; ++st) {
        if (blk == 0) W4_PHASE2()
        const unsigned ldsn = ldsw + st * LSTAGE;
        const int dodma = 3;
        unsigned t0, t1, t2, t3, t4, t5, t6, t7, m0s, stap, xso, wso, sa, sb, sc;
        unsigned long long spair;
        asm volatile("s_mov_b32 %[m0s], m0\n" "s_cmp_gt_i32 %[dodma], 2\n" "s_cselect_b64 vcc, -1, 0\n" W4_WALK
                     PX(0, 0) PX(1, 4096) PX(2, 8192) PX(3, 12288) PX(4, 16384) PX(5, 20480) PX(6, 24576) PX(7, 28672)
                     PW(0, 32768) PW(1, 36864) PW(2, 40960) PW(3, 45056) PW(4, 49152) PW(5, 53248) PW(6, 57344) PW(7, 61440)
                     "s_mov_b32 m0, %[m0s]\n"
                     : W4_TMP_OUT : W4_DMA_IN : "memory", "scc", "vcc");
    }
    // the 256 accumulators are cleared while the first stages are in flight
    float16_t accA[4][2], accB[4][2];                    // rows [wm*128, +64) and [wm*128 + 64, +64)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.f; accB[i][j][r] = 0.f; }
    asm volatile("" : "+a"(accA[0][0]), "+a"(accA[3][1]), "+a"(accB[0][0]), "+a"(accB[3][1]));     // (keeps the clears here)
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // stage 0 of every wave has landed, the epilogue constants are in LDS
    asm volatile("" ::: "memory");

    half8_t w00, w01, w02, w03, x00, x01, x02, x03, w10, w11, w12, w13, x10, x11, x12, x13;
    half8_t w20, w21, w22, w23, x20, x21, x22, x23, w30, w31, w32, w33, x30, x31, x32, x33;
#define RDW(S, A) RD(w##S##0, A, 0) RD(w##S##1, A, 4096) RD(w##S##2, A, 8192) RD(w##S##3, A, 12288)
#define RDX(S, A) RD(x##S##0, A, 0) RD(x##S##1, A, 4096) RD(x##S##2, A, 8192) RD(x##S##3, A, 12288)
    asm volatile(RDW(0, aw0) RDX(0, ax0) RDW(1, aw1) RDX(1, ax1) "s_waitcnt lgkmcnt(0)\n"
                 : [w00] "=&v"(w00), [w01] "=&v"(w01), [w02] "=&v"(w02), [w03] "=&v"(w03), [x00] "=&v"(x00), [x01] "=&v"(x01),
                   [x02] "=&v"(x02), [x03] "=&v"(x03), [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13),
                   [x10] "=&v"(x10), [x11] "=&v"(x11), [x12] "=&v"(x12), [x13] "=&v"(x13)
                 : [aw0] "v"(aw0), [aw1] "v"(aw1), [ax0] "v"(ax0), [ax1] "v"(ax1) : "memory");

    // one MFMA of slice S: accumulator (NI, MI); MI 0-1 live in accA, 2-3 in accB
#define M4(S, NI, E0, E1, E2, E3)                                                                            \
    MF(a##NI##0, w##S##NI, x##S##0) E0 MF(a##NI##1, w##S##NI, x##S##1) E1                                    \
    MF(b##NI##0, w##S##NI, x##S##2) E2 MF(b##NI##1, w##S##NI, x##S##3) E3
#define NO ""
#define TG(R) "v_add_u32 %[" #R "], %[sdel], %[" #R "]\n"
    if (TR) ts[2] = __builtin_amdgcn_s_memtime();
    int sdel = LSTAGE;                                   // + 64 KiB / - 64 KiB: the fragment addresses flip between the two stages
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        // the asm decides "is there a stage ks + 2" itself from the integer nk - ks: a 0 / 1 flag computed here is selected
        // onto the VALU (zero-extended compare -> v_cndmask) and hipcc then hands the asm that VGPR for an "s" operand
        const int dodma = nk - ks;                       // DMA iff > 2
        if (blk == 0) W4_PHASE2()
        const unsigned ldsn = ldsw + cur * LSTAGE;       // stage ks + 2 goes into THIS k-step's buffer (released by barrier A)
        unsigned t0, t1, t2, t3, t4, t5, t6, t7, m0s, stap, xso, wso, sa, sb, sc;
        unsigned long long spair;
        asm volatile(
            "s_mov_b32 %[m0s], m0\n" "s_cmp_gt_i32 %[dodma], 2\n" "s_cselect_b64 vcc, -1, 0\n"
            "s_waitcnt lgkmcnt(0)\n"                     // nothing of the compiler's (SMEM) may be pending below
            // slice 0 (16 MFMAs) + the 16 fragment reads of slices 2 and 3; the scalar walk to stage ks + 2 in three pieces
            M4(0, 0, RD(w20, aw2, 0), RD(w21, aw2, 4096), RD(w22, aw2, 8192), RD(w23, aw2, 12288))
            M4(0, 1, RD(x20, ax2, 0), RD(x21, ax2, 4096), RD(x22, ax2, 8192), RD(x23, ax2, 12288))
            M4(0, 2, RD(w30, aw3, 0) "s_cbranch_vccz .Lk1%=\n" W4_WALK1 ".Lk1%=:\n", RD(w31, aw3, 4096),
                     RD(w32, aw3, 8192) "s_cbranch_vccz .Lk2%=\n" W4_WALK2 ".Lk2%=:\n", RD(w33, aw3, 12288))
            M4(0, 3, RD(x30, ax3, 0) "s_cbranch_vccz .Lk3%=\n" W4_WALK3 ".Lk3%=:\n", RD(x31, ax3, 4096), RD(x32, ax3, 8192), RD(x33, ax3, 12288))
            // slice 1: the fragment addresses flip to the other stage, the effective X offsets of pieces 0-3; all reads of this
            // stage done -> barrier A frees its buffer
            M4(1, 0, TG(aw0) TG(ax0) PXA(0), TG(aw1) TG(ax1) PXA(1), TG(aw2) TG(ax2) PXA(2), TG(aw3) TG(ax3) PXA(3))
            "s_waitcnt lgkmcnt(0)\n" "s_barrier\n"
            M4(1, 1, PXB(0, 0), PXA(4), PXB(1, 4096), PXA(5))
            M4(1, 2, PXB(2, 8192), PXA(6), PXB(3, 12288), PXA(7))
            M4(1, 3, PXB(4, 16384), NO, PXB(5, 20480), NO)
            // slice 2
            M4(2, 0, PXB(6, 24576), NO, PXB(7, 28672), NO)
            M4(2, 1, NO, NO, NO, NO)
            // 8 pieces issued: the 16 of stage ks + 1 (issued one k-step ago) have landed once <= 8 are outstanding
            "s_cbranch_vccz .Lw0%=\n" "s_waitcnt vmcnt(8)\n" "s_branch .Lw1%=\n" ".Lw0%=:\n" "s_waitcnt vmcnt(0)\n" ".Lw1%=:\n"
            "s_barrier\n"
            // rest of slice 2 + slice 3: the 16 fragment reads of slices 0 and 1 of stage ks + 1 and the 8 W pieces
            M4(2, 2, RD(w00, aw0, 0) PW(0, 32768), RD(w01, aw0, 4096), RD(w02, aw0, 8192) PW(1, 36864), RD(w03, aw0, 12288))
            M4(2, 3, RD(x00, ax0, 0) PW(2, 40960), RD(x01, ax0, 4096), RD(x02, ax0, 8192) PW(3, 45056), RD(x03, ax0, 12288))
            M4(3, 0, RD(w10, aw1, 0) PW(4, 49152), RD(w11, aw1, 4096), RD(w12, aw1, 8192) PW(5, 53248), RD(w13, aw1, 12288))
            M4(3, 1, RD(x10, ax1, 0) PW(6, 57344), RD(x11, ax1, 4096), RD(x12, ax1, 8192) PW(7, 61440), RD(x13, ax1, 12288))
            M4(3, 2, NO, NO, NO, NO)
            M4(3, 3, NO, NO, NO, NO)
            "s_waitcnt lgkmcnt(0)\n"
            "s_mov_b32 m0, %[m0s]\n"
            : [a00] "+a"(accA[0][0]), [a01] "+a"(accA[0][1]), [a10] "+a"(accA[1][0]), [a11] "+a"(accA[1][1]),
              [a20] "+a"(accA[2][0]), [a21] "+a"(accA[2][1]), [a30] "+a"(accA[3][0]), [a31] "+a"(accA[3][1]),
              [b00] "+a"(accB[0][0]), [b01] "+a"(accB[0][1]), [b10] "+a"(accB[1][0]), [b11] "+a"(accB[1][1]),
              [b20] "+a"(accB[2][0]), [b21] "+a"(accB[2][1]), [b30] "+a"(accB[3][0]), [b31] "+a"(accB[3][1]),
              [w00] "+v"(w00), [w01] "+v"(w01), [w02] "+v"(w02), [w03] "+v"(w03), [x00] "+v"(x00), [x01] "+v"(x01),
              [x02] "+v"(x02), [x03] "+v"(x03), [w10] "+v"(w10), [w11] "+v"(w11), [w12] "+v"(w12), [w13] "+v"(w13),
              [x10] "+v"(x10), [x11] "+v"(x11), [x12] "+v"(x12), [x13] "+v"(x13),
              [w20] "=&v"(w20), [w21] "=&v"(w21), [w22] "=&v"(w22), [w23] "=&v"(w23), [x20] "=&v"(x20), [x21] "=&v"(x21),
              [x22] "=&v"(x22), [x23] "=&v"(x23), [w30] "=&v"(w30), [w31] "=&v"(w31), [w32] "=&v"(w32), [w33] "=&v"(w33),
              [x30] "=&v"(x30), [x31] "=&v"(x31), [x32] "=&v"(x32), [x33] "=&v"(x33),
              [aw0] "+v"(aw0), [aw1] "+v"(aw1), [aw2] "+v"(aw2), [aw3] "+v"(aw3), [ax0] "+v"(ax0), [ax1] "+v"(ax1),
              [ax2] "+v"(ax2), [ax3] "+v"(ax3), W4_TMP_OUT
            : W4_DMA_IN, [sdel] "s"(sdel)
            : "memory", "scc", "vcc");
        sdel = -sdel;
        cur ^= 1;
    }
#undef RD
#undef MF
#undef PX
#undef PXA
#undef PXB
#undef W4_WALK1
#undef W4_WALK2
#undef W4_WALK3
#undef PW
#undef W4_WALK
#undef W4_WALK_OUT
#undef W4_WALK_IN
#undef W4_PHASE2
#undef W4_DMA_IN
#undef W4_TMP_OUT
#undef RDW
#undef RDX
#undef M4
#undef NO
#undef TG
    // the MFMAs issued last may still be in flight and the compiler cannot see them
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    if (TR) ts[3] = __builtin_amdgcn_s_memtime();
    const unsigned ldsepi = ldsb + 2 * LSTAGE;
    // row-coalesced fp32 epilogues dump the half tile into this wave's quarter of the stage buffers: every wave's fragment reads must be done
    __builtin_amdgcn_s_barrier();
    const unsigned lbuf = ldsb + wave * CO_BYTES;
    conv_w4_epilogue<GNK, HILO>(p, accA, m0 + wm * 128, n0 + wn * 128, l32, hi32, ldsepi + wn * 512,
                          ldsepi + 1024 + (2 * wm) * 1024 + wn * 512, lbuf);
    if (TR) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts[4] = __builtin_amdgcn_s_memtime(); }
    conv_w4_epilogue<GNK, HILO>(p, accB, m0 + wm * 128 + 64, n0 + wn * 128, l32, hi32, ldsepi + wn * 512,
                          ldsepi + 1024 + (2 * wm + 1) * 1024 + wn * 512, lbuf);
    if (TR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[5] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i] = ts[i];
            t[6] = (unsigned long long)nk;
            t[7] = (unsigned long long)__builtin_amdgcn_s_getreg(0xf814);      // HW_REG_XCC_ID etc. (unused)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Short-K kernel (round 5): 1x1 convs / nn.Linear with K = C_in <= a few k-steps (the q / out / proj_in / proj_out
// projections of attention.py:523-564, the 1x1 shortcuts of resnet.py:286-292).  In the 256x256x64 tile these launches
// spend as long in their prologue (two cold DMA stages) and epilogue (fp32 residual in, fp32 rows out: HBM-bound) as in the
// 8 k-steps between them, and with one 128-KiB workgroup per CU nothing runs beside either: 0.15-0.23 of the MFMA peak,
// 60-65 % of the HBM rate the epilogue alone could reach.  Here
//   * the tile is 128(m) x 256(n) per 256-thread workgroup (4 waves as 2 x 2, the SAME 64(m) x 128(n) wave tile, MFMA
//     32x32x16 order and epilogues as conv_gemm256i_kernel), 75 KiB of LDS and <= 256 VGPRs: TWO workgroups per CU, one wave
//     of each on every SIMD, so one workgroup's epilogue / prologue (memory) runs under the other's k-loop (matrix pipe);
//   * K is walked in 32-column stages through a THREE-stage LDS ring (24 KiB each): two stages are in flight while the
//     third is multiplied, one raw s_barrier per stage and a counted vmcnt (never 0 inside the loop);
//   * LDS rows are 64 B: physical 16-B slot s of row r holds logical slot s ^ ((r >> 2) & 3), applied on the DMA source
//     address and on the fragment reads (conflict-free for the four 16-lane groups of ds_read_b128);
//   * no gather arithmetic: row m of the tile IS pixel m (1x1, stride 1); bias and the time-embedding rows of the tile reach
//     LDS as DMA pieces of their own.
// Per-accumulator K order is the same as in the other kernels (ascending k, one MFMA per 16 columns), so results are
// bit-identical to conv_gemm256i_kernel on the launches both accept (tests/test_kernels_gpu.py).
constexpr int SK_BK = 32;
template <int WM, int WN> struct SkGeom {
    static constexpr int TM = 64 * WM, TN = 128 * WN;
    static constexpr int XB = TM * SK_BK * 2, WB = TN * SK_BK * 2, STAGE = XB + WB, NST = 3;
    static constexpr int EPI = (1 + WM) * TN * 4;                  // bias + one time-embedding row per 64-row block
    static constexpr int LDS = NST * STAGE + EPI;
    static constexpr int XP = TM / 64, WP = TN / 64;               // 1-KiB DMA pieces (16 rows x 64 B) per wave and stage
    static_assert(WM * WN == 4 && TN % 256 == 0, "4 waves; bias / row pieces are 256 floats");
};

// V = 0: compiler-scheduled k-step (four read -> lgkmcnt(0) -> 4-MFMA groups per stage); V = 1: the 12 fragment reads and 16
// MFMAs of a stage as ONE asm statement with exact lgkmcnt counts (LDS returns in order): the reads of the second 16-column
// slice fly behind the MFMAs of the first.  Same per-accumulator K order: bit-identical.
template <int WM, int WN, int GNK, int V = 0>
__global__ __launch_bounds__(256, 2) void conv_gemm_sk_kernel(ConvArgs p) {
    using G = SkGeom<WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi32 = lane >> 5, l32 = lane & 31;
    const unsigned n_tiles = p.n_pad / G::TN;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned mt = tile / n_tiles, nt = tile - mt * n_tiles;
    const long long m0 = (long long)mt * G::TM;
    const int n0 = nt * G::TN;

    // DMA role of a lane inside a piece: row lane >> 2, physical slot lane & 3 <- logical slot (lane & 3) ^ ((row >> 2) & 3)
    const int prow = lane >> 2;
    const int slot_log = (lane & 3) ^ ((lane >> 4) & 3);
    int xpix[G::XP];                                     // pixel of this lane's row in X piece wave + 4 j (-1: past M)
#pragma unroll
    for (int j = 0; j < G::XP; ++j) {
        const long long m = m0 + (wave + 4 * j) * 16 + prow;
        xpix[j] = m < p.M ? (int)m : -1;
    }
    const char* wlane = p.w + (((long long)(n0 + wave * 16 + prow)) * p.k_pad + slot_log * 8) * 2;   // W piece wave + 4 j: + j * wstep
    const long long wstep = 64ll * p.k_pad * 2;
    const int nk = p.k_pad / SK_BK;

    auto issue = [&](int buf, int ks) {
        char* sb = smem + buf * G::STAGE;
        const int kc = ks * SK_BK;
        const bool first = kc < p.c1;
        const char* src = first ? p.a1 : p.a2;
        const int cs = first ? p.c1 : p.c2;
        const int coff = (first ? kc : kc - p.c1) + slot_log * 8;
#pragma unroll
        for (int j = 0; j < G::XP; ++j) {
            const int px = xpix[j];
            const int pxs = first ? px : a2_wrap(p, px);
            const char* g = px >= 0 ? src + ((long long)pxs * cs + coff) * 2 : p.zero_page;
            dma16(g, sb + (wave + 4 * j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < G::WP; ++j) dma16(wlane + j * wstep + (long long)kc * 2, sb + G::XB + (wave + 4 * j) * 1024);
    };

    // epilogue constants as DMA pieces of their own (older than every stage piece on the wave's vmcnt): bias[n0 .. n0 + TN),
    // then per 64-row block of the tile the time-embedding row of that block's batch entry
    char* sepi = smem + G::NST * G::STAGE;
    if (wave < G::TN / 256 && p.bias) dma16((const char*)(p.bias + n0 + wave * 256 + lane * 4), sepi + wave * 1024);
    if (p.rowbias) {
#pragma unroll
        for (int q = wave; q < WM * (G::TN / 256); q += 4) {
            const int blk = q / (G::TN / 256), part = q - blk * (G::TN / 256);
            long long mrow = m0 + blk * 64; if (mrow >= p.M) mrow = 0;
            const float* r = p.rowbias + (long long)((int)(mrow / p.rows_per_batch)) * p.rowbias_stride + n0 + part * 256 + lane * 4;
            dma16((const char*)r, sepi + (1 + blk) * G::TN * 4 + part * 1024);
        }
    }
    issue(0, 0);
    if (nk > 1) issue(1, 1);

    const int wn = wave % WN, wm = wave / WN;
    float16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l32 >> 2) & 3;
    const int offW = G::XB + (wn * 128 + l32) * 64, offX = (wm * 64 + l32) * 64;
    const int so0 = ((0 + hi32) ^ sw) << 4, so1 = ((2 + hi32) ^ sw) << 4;

    const unsigned ldsb = (unsigned)(size_t)(lptr_t)smem;
    int cur = 0, nxt = 2;                                // buffer of stage ks / of stage ks + 2
    for (int ks = 0; ks < nk; ++ks) {
        // stage ks has landed once at most the pieces of stage ks + 1 are still outstanding
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::XP + G::WP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // RAW: every wave's pieces of stage ks; WAR: all reads of stage ks - 1 are done
        asm volatile("" ::: "memory");
        if (ks + 2 < nk) issue(nxt, ks + 2);
        if constexpr (V == 1) {
            const unsigned sbs = ldsb + cur * G::STAGE;
            const unsigned aw0 = sbs + offW + so0, aw1 = sbs + offW + so1, ax0 = sbs + offX + so0, ax1 = sbs + offX + so1;
            half8_t w00, w01, w02, w03, x00, x01, w10, w11, w12, w13, x10, x11;
#define RD(D, A, OFF) "ds_read_b128 %[" #D "], %[" #A "] offset:" #OFF "\n"
#define RDSET(S, A, AX) RD(w##S##0, A, 0) RD(x##S##0, AX, 0) RD(x##S##1, AX, 2048) RD(w##S##1, A, 2048) RD(w##S##2, A, 4096) RD(w##S##3, A, 6144)
#define MF(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], %[" #C "]\n"
#define WT(N) "s_waitcnt lgkmcnt(" #N ")\n"
#define MFSET(S, N0, N1, N2, N3, N4)                                                           \
    WT(N0) MF(c00, w##S##0, x##S##0) WT(N1) MF(c01, w##S##0, x##S##1)                          \
    WT(N2) MF(c10, w##S##1, x##S##0) MF(c11, w##S##1, x##S##1)                                 \
    WT(N3) MF(c20, w##S##2, x##S##0) MF(c21, w##S##2, x##S##1)                                 \
    WT(N4) MF(c30, w##S##3, x##S##0) MF(c31, w##S##3, x##S##1)
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"          // nothing of the compiler's (SMEM) may be counted below
                RDSET(0, aw0, ax0) RDSET(1, aw1, ax1)
                MFSET(0, 10, 9, 8, 7, 6)
                MFSET(1, 4, 3, 2, 1, 0)
                : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),
                  [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]),
                  [w00] "=&v"(w00), [w01] "=&v"(w01), [w02] "=&v"(w02), [w03] "=&v"(w03), [x00] "=&v"(x00), [x01] "=&v"(x01),
                  [w10] "=&v"(w10), [w11] "=&v"(w11), [w12] "=&v"(w12), [w13] "=&v"(w13), [x10] "=&v"(x10), [x11] "=&v"(x11)
                : [aw0] "v"(aw0), [aw1] "v"(aw1), [ax0] "v"(ax0), [ax1] "v"(ax1)
                : "memory");
#undef RD
#undef RDSET
#undef MF
#undef WT
#undef MFSET
        } else {
        const char* st = smem + cur * G::STAGE;
        half8_t fw[2][4], fx[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[0][i] = *(const half8_t*)(st + offW + i * 2048 + so0);
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[0][j] = *(const half8_t*)(st + offX + j * 2048 + so0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[1][i] = *(const half8_t*)(st + offW + i * 2048 + so1);
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[1][j] = *(const half8_t*)(st + offX + j * 2048 + so1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk][i], fx[kk][j], acc[i][j], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    // V = 1: the MFMAs issued last may still be in flight and the compiler cannot see them (see conv_gemm256_kernel)
    if constexpr (V == 1) asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    const unsigned ldsepi = (unsigned)(size_t)(lptr_t)sepi;
    conv_epilogue<4, 2, GNK, true, 0>(p, acc, m0 + wm * 64, n0 + wn * 128, l32, hi32, ldsepi + wn * 512,
                                      ldsepi + (1 + wm) * G::TN * 4 + wn * 512);
}

}  // namespace

#ifdef UAV_DEV_W4_ONLY          // development: compile conv_gemm256w_kernel alone (seconds instead of minutes)
extern "C" int uav_conv_gemm_f16(const uav_conv_params* q, void* stream) {
    ConvArgs a = {};
#ifndef UAV_DEV_W4_GNK
#define UAV_DEV_W4_GNK 0        // -DUAV_DEV_W4_GNK=1|2|3: the statistics-reducing instances
#endif
#ifdef UAV_DEV_W4_HILO
    hipLaunchKernelGGL((conv_gemm256w_kernel<0, 0, true>), dim3(1), dim3(256), 2 * LSTAGE + LEPI_BYTES, (hipStream_t)stream, a);
#else
    hipLaunchKernelGGL(conv_gemm256w_kernel<UAV_DEV_W4_GNK>, dim3(1), dim3(256), 2 * LSTAGE + LEPI_BYTES, (hipStream_t)stream, a);
#endif
    return q ? 0 : 1;
}
#else
namespace {
struct ConvEnv { int korder, tile_order, force_tile, dbg, persist, dmav, sk, sk_maxk, w4, w4_mink; };
const ConvEnv& conv_env() {
    // Environment switches (development A/B only) are read once through a thread-safe magic static.
    static const ConvEnv env = [] {
        auto geti = [](const char* k, int d) { const char* e = getenv(k); return e ? atoi(e) : d; };
        {   // only the loops that still have instances: an unknown value used to fall through to the round-1 kernel silently
            const int v = geti("UAV_CONV_DMAV", 6);
            if (v != 1 && v != 6) { fprintf(stderr, "[uav] UAV_CONV_DMAV=%d has no kernel instance (1: round 2-3 loop, 6: rotated k-step); using 6\n", v); setenv("UAV_CONV_DMAV", "6", 1); }
        }
        return ConvEnv{geti("UAV_CONV_KORDER", 1), geti("UAV_CONV_TILE_ORDER", 1), geti("UAV_CONV_TILE", 0),
                       geti("UAV_CONV_DBG", 0), geti("UAV_CONV_PERSIST", 0), geti("UAV_CONV_DMAV", 6),       // 6: rotated k-step (round 4 default); 1: round 2-3 loop
                       geti("UAV_CONV_SK", 0), geti("UAV_CONV_SK_MAXK", 1024),                                // short-K kernel (round 5 candidate, measured neutral: off) for 1x1 launches with K <= SK_MAXK
                       geti("UAV_CONV_W4", 1), geti("UAV_CONV_W4_MINK", 0)};                                // W4 for K = taps x C_in >= MINK.  Same-box clip A/Bs: run 10 (lane-per-row epilogues) 1.000 (8-wave everywhere) -> 1.057 (W4 everywhere) -> 1.070 (from K = 1024); run 25 (fp32 epilogues through LDS) 1.141 (from 1024) -> 1.142 (768) -> 1.152 (512) -> 1.154 (256) -> 1.155 (everywhere): 0                                                               // four-wave 128x128-wave-tile kernel (round 5) instead of conv_gemm256i_kernel
    }();
    return env;
}
// tile selection: the 256x256 kernel needs n_pad % 256 == 0 and enough tiles to fill 256 CUs
bool conv_uses_big_tile(const uav_conv_params* q) {
    const bool small = (q->c1 == 8 && q->c2 == 0);
    const long long M = (long long)q->n_img * q->ho * q->wo;
    const long long grid256 = ((M + LM - 1) / LM) * (q->n_pad / LN);
    const int force_tile = conv_env().force_tile;
    return !small && (q->n_pad % LN == 0) && (force_tile >= 256 || (force_tile != 128 && grid256 >= 224));
}
// n / d for n < 2^31 as umulhi(n, mul) >> sh (d >= 2: s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1, sh = s - 1; d == 1: sh = 32 = "no division")
void conv_magic(unsigned d, unsigned* mul, unsigned* sh) {
    if (d <= 1) { *mul = 0; *sh = 32; return; }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *mul = (unsigned)(((1ull << (31 + s)) / d) + 1ull);
    *sh = s - 1;
}
// Four-wave kernel (conv_gemm256w_kernel): every launch of the big-tile class whose taps fit a 32-bit validity mask and whose
// sources fit a 32-bit buffer range; LayerNorm-fold instances, the nearest-2x gather and the A/B switches stay with the 8-wave kernels.
bool conv_uses_w4(const uav_conv_params* q) {
    const ConvEnv& env = conv_env();
    if (!env.w4 || env.dbg || env.persist || !env.korder || env.dmav != 6 || (q->flags & (UAV_CONV_PERSISTENT | UAV_CONV_NO_W4))) return false;
    if (q->upsample || q->ln_raw_out || q->ln_stat_in || q->kt * q->kh * q->kw > 32 || q->kt > 8 || q->kh > 8 || q->kw > 8) return false;
    if ((long long)q->kt * q->kh * q->kw * (q->c1 + q->c2) < env.w4_mink) return false;      // A/B: short K stays with the 8-wave kernel
    const unsigned long long px = (unsigned long long)q->n_img * q->hi * q->wi;
    const unsigned long long a2px = q->a2_images ? px / 2 : px;
    const unsigned long long halo = 2ull * (((unsigned long long)q->pad_t * q->hi + q->pad_h) * q->wi + q->pad_w) + 16;   // pixels of displacement range
    if ((px + halo) * q->c1 * 2 >= 0xfffffff0ull || (a2px + halo) * q->c2 * 2 >= 0xfffffff0ull) return false;
    return conv_uses_big_tile(q);
}
// UAV_CONV_OUT_HILO (block tails written as the hi | lo fp16 operand pair of their 1x1 consumer): only the row-coalesced fp32
// epilogue of the four-wave kernel stores that form, so the launch must run there and EVERY wave tile must take it (conv_co_kind).
bool conv_hilo_ok(const uav_conv_params* q) {
    if (!(q->flags & UAV_CONV_OUT_F32) || !conv_uses_w4(q)) return false;
    const long long M = (long long)q->n_img * q->ho * q->wo;
    const bool rf32 = q->flags & UAV_CONV_RES_F32;
    if ((q->flags & (UAV_CONV_GELU | UAV_CONV_QUICK_GELU | UAV_CONV_GEGLU)) || !q->bias || (M % 64) || (q->n % 128) || (q->out_stride & 3) ||
        q->out_stride < 2 * q->n || !q->residual || !rf32 || (q->res_stride & 3) || q->rowbias)
        return false;                        // block tails are residual sums on the fp32 stream: the one epilogue form instantiated
    return !q->gn_partials;                  // (the statistics instances do not carry the pair store)
}
// Short-K kernel: 1x1 / stride 1 launches of the big-tile class with K <= UAV_CONV_SK_MAXK and whole 256-column tiles.
bool conv_uses_sk(const uav_conv_params* q) {
    const ConvEnv& env = conv_env();
    if (!env.sk || env.dbg || env.persist || !env.korder || (q->flags & (UAV_CONV_PERSISTENT | UAV_CONV_NO_SHORTK))) return false;
    if (q->kt != 1 || q->kh != 1 || q->kw != 1 || q->stride != 1 || q->upsample || q->pad_t || q->pad_h || q->pad_w) return false;
    if (q->ho != q->hi || q->wo != q->wi || q->out_map_w || q->a2_center_tap) return false;
    if (q->ln_raw_out || q->ln_stat_in) return false;
    if (q->n != q->n_pad || (q->n_pad % 256) || q->k_pad != q->c1 + q->c2 || q->k_pad > env.sk_maxk) return false;
    return conv_uses_big_tile(q);
}
// Fused GroupNorm statistics are produced by the fast epilogues of the 256x256 kernel only: every wave tile (64 rows x
// 128 channels) must lie inside M x N and qualify for a fast path, and a group must not straddle wave tiles.
int conv_gn_cpg_log2(const uav_conv_params* q) {
    // chunks are runs of consecutive output rows — or, for strided output rows (out_map), placed by gn_chunk_* in a shared workspace
    if (q->gn_groups <= 0 || (q->n % q->gn_groups) || (q->out_map_w && q->gn_chunk_cpi <= 0)) return -1;
    const int cpg = q->n / q->gn_groups;
    int cl = -1;
    for (int k = 2; k <= 7; ++k) if (cpg == (1 << k)) cl = k;
    if (cl < 0) return -1;
    const long long M = (long long)q->n_img * q->ho * q->wo;
    if (!conv_uses_big_tile(q) || (M % 64) || (q->n % 128)) return -1;        // (the short-K kernel has the same 64 x 128 wave tiles)
    if (q->flags & (UAV_CONV_GEGLU | UAV_CONV_GELU | UAV_CONV_QUICK_GELU)) return -1;
    const bool of32 = q->flags & UAV_CONV_OUT_F32, rf32 = q->flags & UAV_CONV_RES_F32;
    if (of32) {
        if (!q->bias || (q->out_stride & 3) || (q->residual && (!rf32 || (q->res_stride & 3)))) return -1;
        if (q->rowbias && (q->residual || (q->rows_per_batch % 64))) return -1;
    } else if (rf32) {                                          // fp32 residual, fp16 result
        if (!q->bias || q->rowbias || (q->out_stride & 7) || (q->res_stride & 3)) return -1;
    } else {
        if ((q->out_stride & 7) || (q->residual && (q->res_stride & 7))) return -1;
        if (q->rowbias && (q->rows_per_batch % 64)) return -1;
    }
    return cl;
}
// LayerNorm fold (producer: ln_raw_out + ln_stat_out; consumer: ln_stat_in + ln_colsum): only launches whose EVERY wave tile takes
// the staged fast epilogue of the production 256x256 kernel.
bool conv_ln_ok(const uav_conv_params* q) {
    const long long M = (long long)q->n_img * q->ho * q->wo;
    const ConvEnv& env = conv_env();
    if (!conv_uses_big_tile(q) || (M % 64) || (q->n % 128) || q->out_map_w || q->gn_partials || q->rowbias || !q->bias) return false;
    if (env.dbg || env.persist || (env.dmav != 1 && env.dmav != 6) ||
        (q->flags & (UAV_CONV_PERSISTENT | UAV_CONV_GELU | UAV_CONV_QUICK_GELU)))
        return false;                                      // (the LayerNorm-fold instances themselves run the V = 1 loop)
    const bool of32 = q->flags & UAV_CONV_OUT_F32, rf32 = q->flags & UAV_CONV_RES_F32;
    if (q->ln_raw_out) {                                   // producer: fp32 result (+ fp32 residual), rows of n = out_stride values
        if (!q->ln_stat_out || !of32 || (q->flags & UAV_CONV_GEGLU) || (q->out_stride & 3) || q->out_stride != q->n ||
            (q->residual && (!rf32 || (q->res_stride & 3))))
            return false;
    }
    if (q->ln_stat_in) {                                   // consumer: fp16 result, no residual
        if (!q->ln_colsum || of32 || q->residual || (q->out_stride & 7) || q->ln_chunks <= 0 || q->ln_n <= 0 || q->ln_chunks * 128 != q->ln_n ||
            q->ln_n != q->c1 || q->c2 || q->kt * q->kh * q->kw != 1 || ((q->flags & UAV_CONV_GEGLU) && (q->n % 64)))
            return false;
    }
    return true;
}
}  // namespace

extern "C" int uav_conv_gemm_ln_ok(const uav_conv_params* q) {
    if (!q || q->n_pad <= 0 || (!q->ln_raw_out && !q->ln_stat_in)) return 0;
    return conv_ln_ok(q) ? 1 : 0;
}

extern "C" int uav_conv_gemm_hilo_ok(const uav_conv_params* q) {
    if (!q || q->n_pad <= 0) return 0;
    return conv_hilo_ok(q) ? 1 : 0;
}

extern "C" int uav_conv_gemm_gn_chunk_rows(const uav_conv_params* q) {
    if (!q || q->n_pad <= 0 || q->gn_groups <= 0) return 0;
    return conv_gn_cpg_log2(q) >= 0 ? 64 : 0;
}

extern "C" int uav_conv_gemm_f16(const uav_conv_params* q, void* stream) {
    if (!q || !q->a1 || !q->w || !q->out || !q->zero_page) return UAV_EINVAL;
    const bool small = (q->c1 == 8 && q->c2 == 0);
    if (!small && ((q->c1 % 64) || (q->c2 % 64) || q->c1 <= 0 || q->c2 < 0)) return UAV_ESHAPE;
    if (q->c2 > 0 && !q->a2) return UAV_EINVAL;
    if ((q->n_pad % BN) || (q->k_pad % BK) || q->n <= 0 || q->n > q->n_pad) return UAV_ESHAPE;
    const int ntaps = q->kt * q->kh * q->kw;
    const int cin = q->c1 + q->c2;
    if (ntaps <= 0 || ntaps > 27 * 4) return UAV_ESHAPE;
    if ((long long)ntaps * cin > q->k_pad) return UAV_ESHAPE;
    if (!small && (long long)ntaps * cin != q->k_pad) return UAV_ESHAPE;   // cin%64==0 => exact
    if (q->n % 4) return UAV_ESHAPE;
    if (q->out_stride % 4 || (q->residual && (q->res_stride % 4))) return UAV_EALIGN;
    if (q->rowbias && (q->rows_per_batch <= 0 || (q->rowbias_stride % 4))) return UAV_ESHAPE;
    if ((q->flags & UAV_CONV_GEGLU) && ((q->flags & UAV_CONV_OUT_F32) || q->residual || q->rowbias || (q->n % 64)))
        return UAV_ESHAPE;
    if ((q->flags & UAV_CONV_RES_F32) && !q->residual) return UAV_EINVAL;
    if ((q->flags & UAV_CONV_OUT_HILO) && !conv_hilo_ok(q)) return UAV_ESHAPE;      // ask uav_conv_gemm_hilo_ok() first
    if ((q->flags & (UAV_CONV_GELU | UAV_CONV_QUICK_GELU)) && (q->flags & UAV_CONV_GEGLU)) return UAV_ESHAPE;
    if (q->upsample && (q->stride != 1 || q->ho != 2 * q->hi || q->wo != 2 * q->wi)) return UAV_ESHAPE;
    if (q->t_len <= 0 || q->n_img % q->t_len) return UAV_ESHAPE;
    if (q->ho >= 65536 || q->wo >= 65536) return UAV_ESHAPE;
    ConvArgs a;
    a.a1 = (const char*)q->a1; a.a2 = (const char*)q->a2; a.c1 = q->c1; a.c2 = q->c2;
    a.w = (const char*)q->w; a.bias = q->bias; a.rowbias = q->rowbias;
    a.rows_per_batch = q->rows_per_batch; a.rowbias_stride = q->rowbias_stride;
    a.residual = (const char*)q->residual; a.res_stride = q->res_stride;
    a.out = (char*)q->out; a.out_stride = q->out_stride;
    a.n_img = q->n_img; a.t_len = q->t_len; a.hi = q->hi; a.wi = q->wi; a.ho = q->ho; a.wo = q->wo;
    a.kt = q->kt; a.kh = q->kh; a.kw = q->kw; a.stride = q->stride;
    a.pad_t = q->pad_t; a.pad_h = q->pad_h; a.pad_w = q->pad_w; a.upsample = q->upsample;
    a.n = q->n; a.n_pad = q->n_pad; a.k_pad = q->k_pad; a.out_scale = q->out_scale; a.flags = q->flags;
    a.zero_page = (const char*)q->zero_page;
    a.M = (long long)q->n_img * q->ho * q->wo;
    if (a.M <= 0 || a.M >= (1ll << 31)) return UAV_ESHAPE;
    a.a2_pix = 0;
    if (q->a2_images) {
        if (q->a2_images < 0 || q->c2 <= 0 || q->n_img != 2 * q->a2_images || q->upsample || q->stride != 1) return UAV_ESHAPE;
        a.a2_pix = q->a2_images * q->hi * q->wi;
    }
    a.a2_ctr = 0;
    if (q->a2_center_tap) {
        // the kernels that do not skip (128x128, ablation builds, channel-innermost order) still compute the same sum: the
        // off-centre weight entries of source 2 are zero by contract
        if (q->c2 <= 0 || q->kt != 1 || q->stride != 1 || q->upsample || q->pad_t != 0 || q->pad_h != q->kh / 2 || q->pad_w != q->kw / 2 ||
            !(q->kh & 1) || !(q->kw & 1) || small)
            return UAV_ESHAPE;
        a.a2_ctr = 1;
    }
    a.trace = nullptr;
    a.x1_bytes = a.x2_bytes = 0; a.dv_hw_mul = a.dv_wo_mul = a.dv_t_mul = 0; a.dv_hw_sh = a.dv_wo_sh = a.dv_t_sh = 32;
    a.lnp_raw = nullptr; a.lnp_stat = nullptr; a.lnc_stat = nullptr; a.lnc_colsum = nullptr; a.lnc_chunks = 0; a.lnc_n = 0; a.lnc_eps = 0.f;
    if (q->ln_raw_out || q->ln_stat_in) {
        if (!conv_ln_ok(q)) return UAV_ESHAPE;             // ask uav_conv_gemm_ln_ok() first
        if (q->ln_raw_out) { a.lnp_raw = (char*)q->ln_raw_out; a.lnp_stat = (float*)q->ln_stat_out; }
        if (q->ln_stat_in) {
            a.lnc_stat = q->ln_stat_in; a.lnc_colsum = q->ln_colsum; a.lnc_chunks = q->ln_chunks; a.lnc_n = q->ln_n; a.lnc_eps = q->ln_eps;
        }
    }
    a.omw = 0; a.omsy = 0; a.omsx = 0; a.omoff = 0;
    if (q->out_map_w > 0) {
        if (q->residual || (q->gn_partials && q->gn_chunk_cpi <= 0) || (q->flags & UAV_CONV_GEGLU) || q->out_map_sy < 0 ||
            q->out_map_sx <= 0 || q->out_map_off < 0)
            return UAV_ESHAPE;
        a.omw = q->out_map_w; a.omsy = q->out_map_sy; a.omsx = q->out_map_sx; a.omoff = q->out_map_off;
    } else if (q->out_map_w < 0) return UAV_ESHAPE;
    a.gn_ws = nullptr; a.gn_groups = 0; a.gn_cpg_log2 = 0; a.gn_chunks = 0; a.gn_cpi = 0; a.gn_cstride = 0; a.gn_coff = 0;
    if (q->gn_partials) {
        const int cl = conv_gn_cpg_log2(q);
        if (cl < 0) return UAV_ESHAPE;             // ask uav_conv_gemm_gn_chunk_rows() first
        a.gn_ws = (float*)q->gn_partials; a.gn_groups = q->gn_groups; a.gn_cpg_log2 = cl; a.gn_chunks = a.M / 64;
        if (q->gn_chunk_cpi > 0) {
            const long long k = a.M / 64;
            if ((k % q->gn_chunk_cpi) || q->gn_chunk_off < 0 || q->gn_chunk_off + q->gn_chunk_cpi > q->gn_chunk_stride ||
                (k / q->gn_chunk_cpi) * (long long)q->gn_chunk_stride > q->gn_chunks_total)
                return UAV_ESHAPE;
            a.gn_cpi = q->gn_chunk_cpi; a.gn_cstride = q->gn_chunk_stride; a.gn_coff = q->gn_chunk_off; a.gn_chunks = q->gn_chunks_total;
        }
    }
    // One-time setup.  The dynamic-LDS attribute of the 256x256 kernels and the CU count are PER DEVICE (std::call_once
    // per device index), so a second GPU, or a second host thread driving the library (bench --clips-per-step), never
    // launches before the attribute is in place.
    const ConvEnv& env = conv_env();
    a.korder = env.korder;
    a.tile_order = env.tile_order;
    if (!a.korder) a.a2_ctr = 0;                   // channel-innermost walk (A/B switch): multiply the zeros
    const long long mtiles = (a.M + BM - 1) / BM;
    const long long grid = mtiles * (q->n_pad / BN);
    if (grid >= (1ll << 31)) return UAV_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const long long mtiles256 = (a.M + LM - 1) / LM;
    const long long grid256 = mtiles256 * (q->n_pad / LN);
    const bool big = conv_uses_big_tile(q);
    if (big && conv_uses_sk(q)) {
        using G = SkGeom<2, 2>;
        constexpr int MAXDEV = 64;
        static std::once_flag sk_once[MAXDEV];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return UAV_EINVAL;
        std::call_once(sk_once[dev], [] {
            const void* fns[] = {(const void*)conv_gemm_sk_kernel<2, 2, 0, 0>, (const void*)conv_gemm_sk_kernel<2, 2, 1, 0>,
                                 (const void*)conv_gemm_sk_kernel<2, 2, 2, 0>, (const void*)conv_gemm_sk_kernel<2, 2, 3, 0>,
                                 (const void*)conv_gemm_sk_kernel<2, 2, 0, 1>, (const void*)conv_gemm_sk_kernel<2, 2, 1, 1>,
                                 (const void*)conv_gemm_sk_kernel<2, 2, 2, 1>, (const void*)conv_gemm_sk_kernel<2, 2, 3, 1>};
            for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        });
        const long long gsk = ((a.M + G::TM - 1) / G::TM) * (q->n_pad / G::TN);
        if (gsk >= (1ll << 31)) return UAV_ESHAPE;
        a.ntiles = (unsigned)gsk;
        const int gnm = a.gn_ws ? gn_mode_of(a.gn_cpg_log2) : 0;
#define SK_LAUNCH(GN, VV) hipLaunchKernelGGL((conv_gemm_sk_kernel<2, 2, GN, VV>), dim3((unsigned)gsk), dim3(256), G::LDS, s, a)
        if (env.sk == 2) {                         // UAV_CONV_SK=2: compiler-scheduled k-step (A/B)
            if (gnm == 0) SK_LAUNCH(0, 0); else if (gnm == 1) SK_LAUNCH(1, 0); else if (gnm == 2) SK_LAUNCH(2, 0); else SK_LAUNCH(3, 0);
        } else {
            if (gnm == 0) SK_LAUNCH(0, 1); else if (gnm == 1) SK_LAUNCH(1, 1); else if (gnm == 2) SK_LAUNCH(2, 1); else SK_LAUNCH(3, 1);
        }
#undef SK_LAUNCH
    } else if (big && conv_uses_w4(q)) {
        constexpr int MAXDEV = 64;
        static std::once_flag w4_once[MAXDEV];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return UAV_EINVAL;
        std::call_once(w4_once[dev], [] {
            const void* fns[] = {(const void*)conv_gemm256w_kernel<0>, (const void*)conv_gemm256w_kernel<1>,
                                 (const void*)conv_gemm256w_kernel<2>, (const void*)conv_gemm256w_kernel<3>,
                                 (const void*)conv_gemm256w_kernel<0, 0, true>};
            for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LSTAGE + LEPI_BYTES);
        });
        const unsigned long long px = (unsigned long long)q->n_img * q->hi * q->wi;
        a.x1_bytes = (unsigned)(px * q->c1 * 2);
        a.x2_bytes = (unsigned)((q->a2_images ? px / 2 : px) * q->c2 * 2);
        conv_magic((unsigned)(q->ho * q->wo), &a.dv_hw_mul, &a.dv_hw_sh);
        conv_magic((unsigned)q->wo, &a.dv_wo_mul, &a.dv_wo_sh);
        conv_magic((unsigned)q->t_len, &a.dv_t_mul, &a.dv_t_sh);
        a.ntiles = (unsigned)grid256;
        const int gnm = a.gn_ws ? gn_mode_of(a.gn_cpg_log2) : 0;
        const size_t lds = 2 * LSTAGE + LEPI_BYTES;
        a.trace = nullptr;
        static const bool w4_trace = getenv("UAV_CONV_W4_TRACE") != nullptr;
        if (w4_trace && gnm == 0) {        // development: phase time stamps of every workgroup, printed to stderr (synchronises!)
            static std::once_flag tr_once;
            std::call_once(tr_once, [] { (void)hipFuncSetAttribute((const void*)conv_gemm256w_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LSTAGE + LEPI_BYTES); });
            unsigned long long* tb = nullptr;
            if (hipMalloc((void**)&tb, (size_t)grid256 * 64) != hipSuccess) return UAV_EINVAL;
            a.trace = tb;
            hipLaunchKernelGGL((conv_gemm256w_kernel<0, 1>), dim3((unsigned)grid256), dim3(256), lds, s, a);
            std::vector<unsigned long long> h((size_t)grid256 * 8);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost);
            (void)hipFree(tb);
            double sum[5] = {0, 0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0;
            for (long long i = 0; i < grid256; ++i) {
                for (int k = 0; k < 5; ++k) sum[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]);
                if (h[i * 8] < tmin) tmin = h[i * 8];
                if (h[i * 8 + 5] > tmax) tmax = h[i * 8 + 5];
            }
            fprintf(stderr, "[w4 trace] tiles %lld nk %llu ticks: setup %.0f prologue %.0f loop %.0f (%.1f / k-step) epiA %.0f epiB %.0f | whole launch %llu ticks\n",
                    grid256, h[6], sum[0] / grid256, sum[1] / grid256, sum[2] / grid256, sum[2] / grid256 / (double)h[6], sum[3] / grid256,
                    sum[4] / grid256, tmax - tmin);
            return uav_launch_status();
        }
        if (q->flags & UAV_CONV_OUT_HILO) hipLaunchKernelGGL((conv_gemm256w_kernel<0, 0, true>), dim3((unsigned)grid256), dim3(256), lds, s, a);
        else if (gnm == 0) hipLaunchKernelGGL(conv_gemm256w_kernel<0>, dim3((unsigned)grid256), dim3(256), lds, s, a);
        else if (gnm == 1) hipLaunchKernelGGL(conv_gemm256w_kernel<1>, dim3((unsigned)grid256), dim3(256), lds, s, a);
        else if (gnm == 2) hipLaunchKernelGGL(conv_gemm256w_kernel<2>, dim3((unsigned)grid256), dim3(256), lds, s, a);
        else hipLaunchKernelGGL(conv_gemm256w_kernel<3>, dim3((unsigned)grid256), dim3(256), lds, s, a);
    } else if (big) {
        constexpr int MAXDEV = 64;
        static std::once_flag dev_once[MAXDEV];
        static long long dev_ncu[MAXDEV];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return UAV_EINVAL;
        std::call_once(dev_once[dev], [dev] {
            const void* fns[] = {(const void*)conv_gemm256_kernel<0>, (const void*)conv_gemm256_kernel<1>,
                                 (const void*)conv_gemm256_kernel<2>, (const void*)conv_gemm256_kernel<3>,
                                 (const void*)conv_gemm256_kernel<4>, (const void*)conv_gemm256_kernel<5>,
                                 (const void*)conv_gemm256_kernel<6>, (const void*)conv_gemm256_kernel<0, 1>,
                                 (const void*)conv_gemm256i_kernel<1>, (const void*)conv_gemm256i_kernel<1, 1>,
                                 (const void*)conv_gemm256i_kernel<1, 2>, (const void*)conv_gemm256i_kernel<1, 3>,
                                 (const void*)conv_gemm256i_kernel<1, 0, 1>, (const void*)conv_gemm256i_kernel<1, 0, 2>,
                                 (const void*)conv_gemm256i_kernel<6>, (const void*)conv_gemm256i_kernel<6, 1>,
                                 (const void*)conv_gemm256i_kernel<6, 2>, (const void*)conv_gemm256i_kernel<6, 3>};
            for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LSTAGE + LEPI_BYTES);
            hipDeviceProp_t prop;
            dev_ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        });
        const long long ncu = dev_ncu[dev];
        const int dbg = env.dbg, persist = env.persist;
        a.ntiles = (unsigned)grid256;
        if (a.lnp_raw) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 0, 1>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
        else if (a.lnc_stat) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 0, 2>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
        else if (a.gn_ws && env.dmav == 6) {       // rotated k-step (default), statistics-reducing instances
            const int gnm = gn_mode_of(a.gn_cpg_log2);
            if (gnm == 1) hipLaunchKernelGGL((conv_gemm256i_kernel<6, 1>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
            else if (gnm == 2) hipLaunchKernelGGL((conv_gemm256i_kernel<6, 2>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
            else hipLaunchKernelGGL((conv_gemm256i_kernel<6, 3>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
        } else if (a.gn_ws) {              // statistics-reducing instances of the production kernel (other env A/B switches do not apply)
            const int gnm = gn_mode_of(a.gn_cpg_log2);
            if (gnm == 1) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 1>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
            else if (gnm == 2) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 2>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
            else hipLaunchKernelGGL((conv_gemm256i_kernel<1, 3>), dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
        } else if (dbg == 1) hipLaunchKernelGGL(conv_gemm256_kernel<1>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
        else if (dbg == 2) hipLaunchKernelGGL(conv_gemm256_kernel<2>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
        else if (dbg == 4) hipLaunchKernelGGL(conv_gemm256_kernel<4>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
        else if (dbg == 3) hipLaunchKernelGGL(conv_gemm256_kernel<3>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
        else if (dbg == 5) hipLaunchKernelGGL(conv_gemm256_kernel<5>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
        else if (dbg == 6) hipLaunchKernelGGL(conv_gemm256_kernel<6>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
        else if (env.dmav == 6) hipLaunchKernelGGL(conv_gemm256i_kernel<6>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
        else if (env.dmav == 1) hipLaunchKernelGGL(conv_gemm256i_kernel<1>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE + LEPI_BYTES, s, a);
        else if ((persist || (q->flags & UAV_CONV_PERSISTENT)) && grid256 > ncu) {
            // persistent form: one workgroup per CU walks tiles wg, wg + ncu, ... (UAV_CONV_PERSIST=0 disables)
            hipLaunchKernelGGL((conv_gemm256_kernel<0, 1>), dim3((unsigned)ncu), dim3(512), 2 * LSTAGE, s, a);
        } else hipLaunchKernelGGL(conv_gemm256_kernel<0>, dim3((unsigned)grid256), dim3(512), 2 * LSTAGE, s, a);
    } else if (small)
        hipLaunchKernelGGL(conv_gemm_kernel<1>, dim3((unsigned)grid), dim3(256), 2 * STAGE_BYTES, s, a);
    else
        hipLaunchKernelGGL(conv_gemm_kernel<0>, dim3((unsigned)grid), dim3(256), 2 * STAGE_BYTES, s, a);
    return uav_launch_status();
}
#endif  // UAV_DEV_W4_ONLY
