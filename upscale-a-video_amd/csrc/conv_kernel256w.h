// conv_gemm256w_kernel (template): included by conv_gemm256w.hip (product instances) and conv_gemm_dev.hip (the stamped instance).
#pragma once
#include "conv_common.h"

namespace {
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
}  // namespace
