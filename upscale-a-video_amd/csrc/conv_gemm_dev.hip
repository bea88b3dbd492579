// Development kernels of the implicit-GEMM convolution — NOT part of libuav_hip.so.  Compiled only with -DUAV_DEV_KERNELS
// (tools/ab/build_dev.sh builds tools/ab/libuav_hip_dev.so from ALL sources with that macro; UAV_HIP_LIB selects it):
//   * conv_gemm256_kernel<DBG, PERSIST>      the round-1 256 x 256 kernel, its ablation builds (UAV_CONV_DBG) and the persistent tile walk
//   * conv_gemm256i_kernel<1, GNK, LNF>      the round 2-3 loop (UAV_CONV_DMAV=1) and the LayerNorm-fold producer / consumer instances
//                                            (UAV_LN_FOLD=1: measured slower and outside the parity bar)
//   * conv_gemm_sk_kernel                    the short-K kernel (round 5 candidate, measured neutral; UAV_CONV_SK=1|2)
//   * conv_gemm256w_kernel<0, 1>             the s_memtime-stamped four-wave instance (UAV_CONV_W4_TRACE=1; allocates, synchronises, prints)
// In the product build this file is an empty translation unit.
#ifdef UAV_DEV_KERNELS
#include "conv_kernel256i.h"
#include "conv_kernel256w.h"

namespace {
// ---------------------------------------------------------------------------------------------
// Large-tile variant: 256(m) x 256(n) x 64(k) per 512-thread workgroup (8 waves; wave tile
// 128(n) x 64(m) = 4x2 MFMA 32x32x16 tiles, 128 fp32 accumulators/lane), two 64-KiB LDS stages,
// one workgroup per CU.  Compared with the 128x128 kernel each wave issues 2x the MFMAs per
// global_load_lds instruction (4:1) and 0.75 ds_read_b128 per MFMA instead of 1, and there are
// 2x the MFMAs between two barriers.  Because only 2 waves share a SIMD, latency is hidden INSIDE
// the wave: fragments are double-buffered in registers (the reads of k-slice kk+1 are issued
// before the MFMAs of slice kk) and the DMA of the next stage is issued in the first two slices.

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

int conv_launch_dev(int which, const ConvArgs& a_in, long long grid, int gn_mode, int lnf, int dbg, int persist, int sk_variant, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int MAXDEV = 64;
    static std::once_flag once[MAXDEV];
    static long long dev_ncu[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return UAV_EINVAL;
    using G = SkGeom<2, 2>;
    std::call_once(once[dev], [dev] {
        const void* fns[] = {(const void*)conv_gemm256_kernel<0>, (const void*)conv_gemm256_kernel<1>, (const void*)conv_gemm256_kernel<2>,
                             (const void*)conv_gemm256_kernel<3>, (const void*)conv_gemm256_kernel<4>, (const void*)conv_gemm256_kernel<5>,
                             (const void*)conv_gemm256_kernel<6>, (const void*)conv_gemm256_kernel<0, 1>,
                             (const void*)conv_gemm256i_kernel<1>, (const void*)conv_gemm256i_kernel<1, 1>, (const void*)conv_gemm256i_kernel<1, 2>,
                             (const void*)conv_gemm256i_kernel<1, 3>, (const void*)conv_gemm256i_kernel<1, 0, 1>, (const void*)conv_gemm256i_kernel<1, 0, 2>,
                             (const void*)conv_gemm256w_kernel<0, 1>};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LSTAGE + LEPI_BYTES);
        const void* sks[] = {(const void*)conv_gemm_sk_kernel<2, 2, 0, 0>, (const void*)conv_gemm_sk_kernel<2, 2, 1, 0>,
                             (const void*)conv_gemm_sk_kernel<2, 2, 2, 0>, (const void*)conv_gemm_sk_kernel<2, 2, 3, 0>,
                             (const void*)conv_gemm_sk_kernel<2, 2, 0, 1>, (const void*)conv_gemm_sk_kernel<2, 2, 1, 1>,
                             (const void*)conv_gemm_sk_kernel<2, 2, 2, 1>, (const void*)conv_gemm_sk_kernel<2, 2, 3, 1>};
        for (const void* f : sks) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        hipDeviceProp_t prop;
        dev_ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    });
    const unsigned g = (unsigned)grid;
    const size_t ldsi = 2 * LSTAGE + LEPI_BYTES;
    if (which == 2) {                              // short-K kernel
#define SK_LAUNCH(GN, VV) hipLaunchKernelGGL((conv_gemm_sk_kernel<2, 2, GN, VV>), dim3(g), dim3(256), G::LDS, s, a)
        if (sk_variant == 2) {                     // compiler-scheduled k-step (A/B)
            if (gn_mode == 0) SK_LAUNCH(0, 0); else if (gn_mode == 1) SK_LAUNCH(1, 0); else if (gn_mode == 2) SK_LAUNCH(2, 0); else SK_LAUNCH(3, 0);
        } else {
            if (gn_mode == 0) SK_LAUNCH(0, 1); else if (gn_mode == 1) SK_LAUNCH(1, 1); else if (gn_mode == 2) SK_LAUNCH(2, 1); else SK_LAUNCH(3, 1);
        }
#undef SK_LAUNCH
        return uav_launch_status();
    }
    if (which == 3) {                              // phase time stamps of every workgroup of the four-wave kernel, printed to stderr
        unsigned long long* tb = nullptr;
        if (hipMalloc((void**)&tb, (size_t)grid * 64) != hipSuccess) return UAV_EINVAL;
        a.trace = tb;
        hipLaunchKernelGGL((conv_gemm256w_kernel<0, 1>), dim3(g), dim3(256), ldsi, s, a);
        std::vector<unsigned long long> h((size_t)grid * 8);
        (void)hipStreamSynchronize(s);
        const hipError_t ce = hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(tb);
        if (ce != hipSuccess) return (int)ce;
        double sum[5] = {0, 0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0;
        for (long long i = 0; i < grid; ++i) {
            for (int k = 0; k < 5; ++k) sum[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]);
            if (h[i * 8] < tmin) tmin = h[i * 8];
            if (h[i * 8 + 5] > tmax) tmax = h[i * 8 + 5];
        }
        fprintf(stderr, "[w4 trace] tiles %lld nk %llu ticks: setup %.0f prologue %.0f loop %.0f (%.1f / k-step) epiA %.0f epiB %.0f | whole launch %llu ticks\n",
                grid, h[6], sum[0] / grid, sum[1] / grid, sum[2] / grid, sum[2] / grid / (double)h[6], sum[3] / grid, sum[4] / grid, tmax - tmin);
        return uav_launch_status();
    }
    if (which == 1) {                              // round 2-3 loop (V = 1) and its LayerNorm-fold / statistics instances
        if (lnf == 1) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 0, 1>), dim3(g), dim3(512), ldsi, s, a);
        else if (lnf == 2) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 0, 2>), dim3(g), dim3(512), ldsi, s, a);
        else if (gn_mode == 1) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 1>), dim3(g), dim3(512), ldsi, s, a);
        else if (gn_mode == 2) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 2>), dim3(g), dim3(512), ldsi, s, a);
        else if (gn_mode == 3) hipLaunchKernelGGL((conv_gemm256i_kernel<1, 3>), dim3(g), dim3(512), ldsi, s, a);
        else hipLaunchKernelGGL(conv_gemm256i_kernel<1>, dim3(g), dim3(512), ldsi, s, a);
        return uav_launch_status();
    }
    // round-1 kernel: ablation builds, the persistent walk, the plain form
    const long long ncu = dev_ncu[dev];
    if (dbg == 1) hipLaunchKernelGGL(conv_gemm256_kernel<1>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    else if (dbg == 2) hipLaunchKernelGGL(conv_gemm256_kernel<2>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    else if (dbg == 3) hipLaunchKernelGGL(conv_gemm256_kernel<3>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    else if (dbg == 4) hipLaunchKernelGGL(conv_gemm256_kernel<4>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    else if (dbg == 5) hipLaunchKernelGGL(conv_gemm256_kernel<5>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    else if (dbg == 6) hipLaunchKernelGGL(conv_gemm256_kernel<6>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    else if (persist && grid > ncu) hipLaunchKernelGGL((conv_gemm256_kernel<0, 1>), dim3((unsigned)ncu), dim3(512), 2 * LSTAGE, s, a);
    else hipLaunchKernelGGL(conv_gemm256_kernel<0>, dim3(g), dim3(512), 2 * LSTAGE, s, a);
    return uav_launch_status();
}
#endif  // UAV_DEV_KERNELS
