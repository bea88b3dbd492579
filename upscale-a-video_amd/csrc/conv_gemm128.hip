// 128 x 128 x 64 tile kernel of the implicit-GEMM convolution (see conv_common.h for the family map).
#include "conv_common.h"

namespace {
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
}  // namespace

int conv_launch_tile128(const ConvArgs& a, long long grid, bool small, hipStream_t s) {
    if (small) hipLaunchKernelGGL(conv_gemm_kernel<1>, dim3((unsigned)grid), dim3(256), 2 * STAGE_BYTES, s, a);
    else hipLaunchKernelGGL(conv_gemm_kernel<0>, dim3((unsigned)grid), dim3(256), 2 * STAGE_BYTES, s, a);
    return uav_launch_status();
}
