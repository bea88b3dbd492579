// conv_gemm256i_kernel (template): included by conv_gemm256i.hip (product instances) and conv_gemm_dev.hip (development instances).
#pragma once
#include "conv_common.h"

namespace {
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
}  // namespace
