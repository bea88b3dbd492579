// K11a — fp32 implicit-GEMM convolution / linear for gfx950: the RAFT optical-flow network
// (reference models_video/RAFT/{extractor,update,corr}.py) runs in fp32 in the reference
// (raft_bi.py:26 mixed_precision=False), so its convolutions use the exact-fp32 MFMA
// v_mfma_f32_32x32x2_f32 (bit-identical to an fmaf chain, 157 TFLOP/s peak = 1/16 of fp16).
//
// Same structure as the fp16 128x128 kernel of conv_gemm.hip (swapped MFMA: lane = pixel,
// global_load_lds DMA gather with source-side swizzle, zero page for padding, two LDS stages) with
// 4-byte elements: a 16-B slot holds 4 channels, a k-step covers 32 channels (128-B LDS rows).
// Serves 7x7/s2, 3x3, 1x1, 1x5, 5x1 convs, the all-pairs correlation (a "linear" whose weight
// matrix is the second feature map) and two-source channel concats.  Epilogue: bias, residual,
// scale, ReLU / sigmoid / tanh.  Algorithmic FLOP per launch = 2*M*N*taps*C_in.
#include "uav_common.h"

namespace {

constexpr int BM = 128, BN = 128, BKE = 32;            // tile; k-step in elements
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;  // 128-B rows
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;

struct ConvArgsF {
    const char* a1; const char* a2; int c1, c2;
    const char* w; const float* bias;
    const char* residual; int res_stride;
    char* out; int out_stride;
    int n_img, t_len, hi, wi, ho, wo, kt, kh, kw, stride, pad_t, pad_h, pad_w, upsample;
    int n, n_pad, k_pad; float out_scale; unsigned flags;
    const char* zero_page;
    long long M;
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

UAV_DEVINL void dma16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

UAV_DEVINL int src_pixel(const ConvArgsF& p, int img, int tloc, int yo, int xo, int dt, int dy, int dx) {
    int tt = tloc + dt - p.pad_t;
    bool ok = (tt >= 0) & (tt < p.t_len);
    int yi = yo * p.stride + dy - p.pad_h, xi = xo * p.stride + dx - p.pad_w;
    ok = ok & (yi >= 0) & (yi < p.hi) & (xi >= 0) & (xi < p.wi);
    int pix = ((img + dt - p.pad_t) * p.hi + yi) * p.wi + xi;
    return ok ? pix : -1;
}

template <int SMALL>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_kernel(ConvArgsF p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi32 = lane >> 5, l32 = lane & 31;
    const unsigned n_tiles = p.n_pad / BN;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned mt = bid / n_tiles, nt = bid - mt * n_tiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    const int slot_log = (tid & 7) ^ ((tid >> 4) & 7);
    const int rbase = tid >> 3;
    int img[4], tloc[4], yx[4];
    bool mval[4];
    const int hw_o = p.ho * p.wo;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        long long m = m0 + ps * 32 + rbase;
        mval[ps] = m < p.M;
        int mm = mval[ps] ? (int)m : 0;
        int im = mm / hw_o; int rem = mm - im * hw_o;
        int yo = rem / p.wo; int xo = rem - yo * p.wo;
        img[ps] = im; tloc[ps] = im % p.t_len; yx[ps] = (yo << 16) | xo;
    }
    const int cin = p.c1 + p.c2;
    const int khw = p.kh * p.kw, ntaps = p.kt * khw;
    const int nk = p.k_pad / BKE;
    const char* wrow = p.w + ((long long)(n0 + rbase) * p.k_pad + slot_log * 4) * 4;
    int pix[4] = {-1, -1, -1, -1};
    int nxt_tap = 0, nxt_c = 0;

    auto issue = [&](int stage, int ks) {
        char* sA = smem + stage * STAGE_BYTES;
        char* sB = sA + A_BYTES;
        if (SMALL) {                                        // cin_p == 4: one 16-B slot = one tap of one pixel
            int tap = ks * 8 + slot_log;
            bool tok = tap < ntaps;
            int dt = tap / khw; int rem = tap - dt * khw; int dy = rem / p.kw; int dx = rem - dy * p.kw;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                int px = (tok && mval[ps]) ? src_pixel(p, img[ps], tloc[ps], yx[ps] >> 16, yx[ps] & 0xffff, dt, dy, dx) : -1;
                const char* g = px >= 0 ? p.a1 + (long long)px * 16 : p.zero_page;
                dma16(g, sA + (ps * 256 + wave * 64) * 16);
            }
        } else {
            if (nxt_c == 0) {
                int dt = nxt_tap / khw; int rem = nxt_tap - dt * khw; int dy = rem / p.kw; int dx = rem - dy * p.kw;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps)
                    pix[ps] = mval[ps] ? src_pixel(p, img[ps], tloc[ps], yx[ps] >> 16, yx[ps] & 0xffff, dt, dy, dx) : -1;
            }
            const bool first = nxt_c < p.c1;
            const char* src = first ? p.a1 : p.a2;
            const int cs = first ? p.c1 : p.c2;
            const int coff = (first ? nxt_c : nxt_c - p.c1) + slot_log * 4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const char* g = pix[ps] >= 0 ? src + ((long long)pix[ps] * cs + coff) * 4 : p.zero_page;
                dma16(g, sA + (ps * 256 + wave * 64) * 16);
            }
            nxt_c += BKE;
            if (nxt_c >= cin) { nxt_c = 0; ++nxt_tap; }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
            dma16(wrow + ((long long)ps * 32 * p.k_pad + (long long)ks * BKE) * 4, sB + (ps * 256 + wave * 64) * 16);
    };

    const int wn = wave & 1, wm = wave >> 1;
    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sw = (l32 >> 1) & 7;
    const int offW = A_BYTES + (wn * 64 + l32) * 128, offX = (wm * 64 + l32) * 128;

    issue(0, 0);
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ks + 1 < nk) issue(cur ^ 1, ks + 1);
        const char* st = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int so = ((kk * 2 + hi32) ^ sw) << 4;
            float4_t w0 = *(const float4_t*)(st + offW + so), w1 = *(const float4_t*)(st + offW + 4096 + so);
            float4_t x0 = *(const float4_t*)(st + offX + so), x1 = *(const float4_t*)(st + offX + 4096 + so);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[j], x0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[j], x1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[j], x0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[j], x1[j], acc[1][1], 0, 0, 0);
            }
        }
        cur ^= 1;
    }

    const int act = (p.flags >> 2) & 7;                    // 1 relu, 2 sigmoid, 4 tanh
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const long long m = m0 + wm * 64 + mi * 32 + l32;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * hi32;
                if (n >= p.n) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
                if (p.bias) {
                    float4_t b = *(const float4_t*)(p.bias + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += b[j];
                }
                if (p.residual) {
                    float4_t r = *(const float4_t*)(p.residual + ((long long)m * p.res_stride + n) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += r[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = v[j] * p.out_scale;
                    if (act == 1) t = fmaxf(t, 0.f);
                    else if (act == 2) t = 1.0f / (1.0f + expf(-t));
                    else if (act == 4) t = tanhf(t);
                    v[j] = t;
                }
                float4_t o = {v[0], v[1], v[2], v[3]};
                *(float4_t*)(p.out + ((long long)m * p.out_stride + n) * 4) = o;
            }
    }
}

}  // namespace

extern "C" int uav_conv_gemm_f32(const uav_conv_params* q, void* stream) {
    if (!q || !q->a1 || !q->w || !q->out || !q->zero_page) return UAV_EINVAL;
    const bool small = (q->c1 == 4 && q->c2 == 0);
    if (!small && ((q->c1 % 32) || (q->c2 % 32) || q->c1 <= 0 || q->c2 < 0)) return UAV_ESHAPE;
    if (q->c2 > 0 && !q->a2) return UAV_EINVAL;
    if ((q->n_pad % BN) || (q->k_pad % BKE) || q->n <= 0 || q->n > q->n_pad || (q->n % 4)) return UAV_ESHAPE;
    const int ntaps = q->kt * q->kh * q->kw, cin = q->c1 + q->c2;
    if (ntaps <= 0 || (long long)ntaps * cin > q->k_pad) return UAV_ESHAPE;
    if (!small && (long long)ntaps * cin != q->k_pad) return UAV_ESHAPE;
    if ((q->out_stride % 4) || (q->residual && (q->res_stride % 4))) return UAV_EALIGN;
    if (q->upsample || q->rowbias || (q->flags & (UAV_CONV_GEGLU))) return UAV_ESHAPE;
    if (q->t_len <= 0 || q->n_img % q->t_len || q->ho >= 65536 || q->wo >= 65536) return UAV_ESHAPE;
    ConvArgsF a;
    a.a1 = (const char*)q->a1; a.a2 = (const char*)q->a2; a.c1 = q->c1; a.c2 = q->c2;
    a.w = (const char*)q->w; a.bias = q->bias; a.residual = (const char*)q->residual; a.res_stride = q->res_stride;
    a.out = (char*)q->out; a.out_stride = q->out_stride;
    a.n_img = q->n_img; a.t_len = q->t_len; a.hi = q->hi; a.wi = q->wi; a.ho = q->ho; a.wo = q->wo;
    a.kt = q->kt; a.kh = q->kh; a.kw = q->kw; a.stride = q->stride; a.pad_t = q->pad_t; a.pad_h = q->pad_h; a.pad_w = q->pad_w;
    a.upsample = 0; a.n = q->n; a.n_pad = q->n_pad; a.k_pad = q->k_pad; a.out_scale = q->out_scale; a.flags = q->flags;
    a.zero_page = (const char*)q->zero_page;
    a.M = (long long)q->n_img * q->ho * q->wo;
    if (a.M <= 0 || a.M >= (1ll << 31)) return UAV_ESHAPE;
    const long long grid = ((a.M + BM - 1) / BM) * (q->n_pad / BN);
    hipStream_t s = (hipStream_t)stream;
    if (small) hipLaunchKernelGGL(conv_gemm_f32_kernel<1>, dim3((unsigned)grid), dim3(256), 2 * STAGE_BYTES, s, a);
    else hipLaunchKernelGGL(conv_gemm_f32_kernel<0>, dim3((unsigned)grid), dim3(256), 2 * STAGE_BYTES, s, a);
    return uav_launch_status();
}
