// K12 — colour-correction post-process of the CLI on the decoded frames (gfx950), fp32 planes (T*C, H, W).
//
// Replaces (reference models_video/color_correction.py, inference_upscale_a_video.py:322-333):
//   F.interpolate(vframes, scale_factor=4, mode='bicubic')          uav_resize_bicubic_f32
//   adaptive_instance_normalization (:59-71, calc_mean_std :43-57)   uav_plane_stats_f32 + uav_adain_apply_f32
//   wavelet_blur / wavelet_decomposition / _reconstruction (:73-118) uav_atrous_blur_f32 (+ uav_axpby_f32)
//
// All of it is HBM-bound elementwise / small-stencil work on 1280x1280 frames (20 MB per frame and channel triple):
// coalesced 4-B accesses along W, the stencil's 9 taps (dilation up to 16) come from L2.  Bounds: 8 B/pixel for the
// blur (read + write; +8 with the fused high-frequency accumulation), 4 B/pixel for the statistics pass, 8 B/pixel for
// the AdaIN apply, 4.25 B/pixel for the 4x bicubic resize.  Statistics are deterministic (two-stage, fp64 combine).
#include "uav_common.h"

namespace {

constexpr int CF_CHUNKS = 64;

__global__ __launch_bounds__(256) void plane_partial_kernel(const float* __restrict__ x, long long hw, double* __restrict__ ws) {
    __shared__ double rs[256], rq[256];
    const int plane = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const long long per = (hw + CF_CHUNKS - 1) / CF_CHUNKS;
    const long long i0 = (long long)chunk * per;
    long long i1 = i0 + per; if (i1 > hw) i1 = hw;
    const float* p = x + (long long)plane * hw;
    double s = 0.0, q = 0.0;
    // fp32 running sums over short strips (<= 64 elements), folded into fp64: keeps the 1.6 M-element reduction exact
    // to ~1e-7 relative without paying fp64 per element
    for (long long i = i0 + tid; i < i1; i += 256 * 64) {
        float fs = 0.f, fq = 0.f;
#pragma unroll 8
        for (int u = 0; u < 64; ++u) {
            const long long j = i + (long long)u * 256;
            if (j < i1) { const float v = p[j]; fs += v; fq += v * v; }
        }
        s += fs; q += fq;
    }
    rs[tid] = s; rq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { ws[((long long)plane * CF_CHUNKS + chunk) * 2] = rs[0]; ws[((long long)plane * CF_CHUNKS + chunk) * 2 + 1] = rq[0]; }
}

// mean and UNBIASED variance per plane (torch.var default, color_correction.py:54)
__global__ void plane_finalize_kernel(const double* __restrict__ ws, int planes, long long hw, float* __restrict__ mean,
                                      float* __restrict__ var) {
    const int plane = blockIdx.x * blockDim.x + threadIdx.x;
    if (plane >= planes) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < CF_CHUNKS; ++c) { s += ws[((long long)plane * CF_CHUNKS + c) * 2]; q += ws[((long long)plane * CF_CHUNKS + c) * 2 + 1]; }
    const double n = (double)hw, m = s / n;
    double v = hw > 1 ? (q - n * m * m) / (n - 1.0) : 0.0;
    if (v < 0.0) v = 0.0;
    mean[plane] = (float)m; var[plane] = (float)v;
}

// out = (x - cm) / sqrt(cv + eps) * sqrt(sv + eps) + sm   (color_correction.py:69-71), 4 elements per thread
__global__ __launch_bounds__(256) void adain_apply_kernel(const float* __restrict__ x, float* __restrict__ out, long long hw,
                                                          const float* __restrict__ cm, const float* __restrict__ cv,
                                                          const float* __restrict__ sm, const float* __restrict__ sv, float eps) {
    const int plane = blockIdx.y;
    const float m = cm[plane], cs = sqrtf(cv[plane] + eps), ss = sqrtf(sv[plane] + eps), smn = sm[plane];
    const long long base = (long long)plane * hw;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= hw && !(base & 3)) {
        float4_t v = *(const float4_t*)(x + base + i), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[j] - m) / cs * ss + smn;
        *(float4_t*)(out + base + i) = o;
    } else {
        for (long long j = i; j < hw && j < i + 4; ++j) out[base + j] = (x[base + j] - m) / cs * ss + smn;
    }
}

// 3x3 a-trous blur [1 2 1]^T [1 2 1] / 16 with dilation `r` on a replicate-padded image (wavelet_blur :73-91) and,
// optionally, the decomposition's running high-frequency sum high += image - low (:101-103).
__global__ __launch_bounds__(256) void atrous_blur_kernel(const float* __restrict__ src, float* __restrict__ low,
                                                          float* __restrict__ high, int h, int w, int r) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const long long base = (long long)blockIdx.z * h * w;
    const int xm = x - r < 0 ? 0 : x - r, xp = x + r > w - 1 ? w - 1 : x + r;
    const int ym = y - r < 0 ? 0 : y - r, yp = y + r > h - 1 ? h - 1 : y + r;
    const float* p = src + base;
    const float* r0 = p + (long long)ym * w; const float* r1 = p + (long long)y * w; const float* r2 = p + (long long)yp * w;
    const float c = r1[x];
    // same tap order as a 3x3 cross-correlation (row-major), weights exact in binary
    float a = 0.0625f * r0[xm];
    a += 0.125f * r0[x]; a += 0.0625f * r0[xp];
    a += 0.125f * r1[xm]; a += 0.25f * c; a += 0.125f * r1[xp];
    a += 0.0625f * r2[xm]; a += 0.125f * r2[x]; a += 0.0625f * r2[xp];
    const long long o = base + (long long)y * w + x;
    low[o] = a;
    if (high) high[o] += c - a;
}

// F.interpolate(mode='bicubic', align_corners=False) with an integer or fractional scale: ATen upsample_bicubic2d
// (cubic convolution, A = -0.75, source index scale*(dst+0.5)-0.5, clamped taps).
UAV_DEVINL float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
UAV_DEVINL float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
UAV_DEVINL void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    c[0] = cc2(t + 1.f, A); c[1] = cc1(t, A); c[2] = cc1(1.f - t, A); c[3] = cc2(2.f - t, A);
}

__global__ __launch_bounds__(256) void resize_bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int hi, int wi,
                                                             int ho, int wo, float sy, float sx) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= wo) return;
    const float* p = src + (long long)blockIdx.z * hi * wi;
    const float ry = sy * (y + 0.5f) - 0.5f, rx = sx * (x + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    float cy[4], cx[4];
    cubic_coeffs(ry - fy, cy); cubic_coeffs(rx - fx, cx);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int yy = iy - 1 + i; yy = yy < 0 ? 0 : (yy > hi - 1 ? hi - 1 : yy);
        const float* row = p + (long long)yy * wi;
        float rsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int xx = ix - 1 + j; xx = xx < 0 ? 0 : (xx > wi - 1 ? wi - 1 : xx);
            rsum += row[xx] * cx[j];
        }
        acc += rsum * cy[i];
    }
    dst[((long long)blockIdx.z * ho + y) * wo + x] = acc;
}

// F.interpolate(mode='area') = adaptive average pooling: window [floor(o*in/out), ceil((o+1)*in/out)) per axis; `mul`
// folds the flow rescale of Propagation.forward (propagation_module.py:206-209: flows * w / w_f).
__global__ __launch_bounds__(256) void resize_area_kernel(const float* __restrict__ src, float* __restrict__ dst, int hi, int wi,
                                                          int ho, int wo, float mul) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= wo) return;
    const float* p = src + (long long)blockIdx.z * hi * wi;
    const int y0 = (int)(((long long)y * hi) / ho), y1 = (int)((((long long)(y + 1)) * hi + ho - 1) / ho);
    const int x0 = (int)(((long long)x * wi) / wo), x1 = (int)((((long long)(x + 1)) * wi + wo - 1) / wo);
    float acc = 0.f;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) acc += p[(long long)yy * wi + xx];
    dst[((long long)blockIdx.z * ho + y) * wo + x] = acc / (float)((y1 - y0) * (x1 - x0)) * mul;
}

}  // namespace

extern "C" int uav_resize_area_f32(const float* src, float* dst, int32_t planes, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                                   float mul, void* stream) {
    if (!src || !dst) return UAV_EINVAL;
    if (planes <= 0 || planes > 65535 || hi <= 0 || wi <= 0 || ho <= 0 || ho > 65535 || wo <= 0) return UAV_ESHAPE;
    hipLaunchKernelGGL(resize_area_kernel, dim3((wo + 255) / 256, ho, planes), dim3(256), 0, (hipStream_t)stream, src, dst, hi, wi,
                       ho, wo, mul);
    return uav_launch_status();
}

// ---- K13: frame I/O conversions of the CLI, HBM-bound elementwise -----------------------------------
namespace {
// (T,C,H,W) frames in 0..255 (uint8, or the fp32 tensor `read_frame_from_videos` returns) -> (C,T,H,W) fp32 in [-1,1]:
// `(vframes / 255. - 0.5) * 2` then 'b t c h w -> b c t h w' (inference_upscale_a_video.py:180,186-187), same op order.
template <typename T>
__global__ __launch_bounds__(256) void frames_to_clip_kernel(const T* __restrict__ src, float* __restrict__ dst, int t_len, int c,
                                                             long long hw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const int tc = blockIdx.y, t = tc / c, ch = tc - t * c;
    const float v = (float)src[(long long)tc * hw + i];
    dst[((long long)ch * t_len + t) * hw + i] = (v / 255.0f - 0.5f) * 2.0f;
}
// (T,C,H,W) fp32 in [-1,1] -> (T,H,W,C) uint8: `(output / 2 + 0.5).clamp(0, 1) * 255`, 't c h w -> t h w c', numpy
// `.astype(np.uint8)` = truncation (inference_upscale_a_video.py:357-359).  One thread per pixel, C <= 4 channels.
__global__ __launch_bounds__(256) void clip_to_u8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int c,
                                                         long long hw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const int t = blockIdx.y;
    for (int ch = 0; ch < c; ++ch) {
        float v = src[((long long)t * c + ch) * hw + i] / 2.0f + 0.5f;
        v = fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f;
        dst[((long long)t * hw + i) * c + ch] = (unsigned char)(int)v;
    }
}
}  // namespace

extern "C" int uav_frames_to_clip_f32(const void* frames_tchw, int32_t src_is_u8, float* clip_cthw, int32_t t_len, int32_t c,
                                      int64_t hw, void* stream) {
    if (!frames_tchw || !clip_cthw) return UAV_EINVAL;
    if (t_len <= 0 || c <= 0 || hw <= 0 || (int64_t)t_len * c > 65535) return UAV_ESHAPE;
    const dim3 grid((unsigned)((hw + 255) / 256), (unsigned)(t_len * c));
    if (src_is_u8)
        hipLaunchKernelGGL(frames_to_clip_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)frames_tchw,
                           clip_cthw, t_len, c, (long long)hw);
    else
        hipLaunchKernelGGL(frames_to_clip_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)frames_tchw, clip_cthw,
                           t_len, c, (long long)hw);
    return uav_launch_status();
}

extern "C" int uav_clip_to_frames_u8(const float* frames_tchw, void* frames_thwc_u8, int32_t t_len, int32_t c, int64_t hw,
                                     void* stream) {
    if (!frames_tchw || !frames_thwc_u8) return UAV_EINVAL;
    if (t_len <= 0 || t_len > 65535 || c <= 0 || c > 4 || hw <= 0) return UAV_ESHAPE;
    hipLaunchKernelGGL(clip_to_u8_kernel, dim3((unsigned)((hw + 255) / 256), (unsigned)t_len), dim3(256), 0, (hipStream_t)stream,
                       frames_tchw, (unsigned char*)frames_thwc_u8, c, (long long)hw);
    return uav_launch_status();
}

extern "C" int64_t uav_plane_stats_workspace_bytes(int32_t planes) { return (int64_t)planes * CF_CHUNKS * 2 * 8; }

extern "C" int uav_plane_stats_f32(const float* x, int32_t planes, int64_t hw, float* mean_out, float* var_out, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    if (!x || !mean_out || !var_out || !workspace) return UAV_EINVAL;
    if (planes <= 0 || planes > 65535 || hw <= 0) return UAV_ESHAPE;
    if (workspace_bytes < uav_plane_stats_workspace_bytes(planes)) return UAV_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(plane_partial_kernel, dim3(CF_CHUNKS, planes), dim3(256), 0, s, x, (long long)hw, (double*)workspace);
    hipLaunchKernelGGL(plane_finalize_kernel, dim3((planes + 63) / 64), dim3(64), 0, s, (const double*)workspace, planes,
                       (long long)hw, mean_out, var_out);
    return uav_launch_status();
}

extern "C" int uav_adain_apply_f32(const float* x, float* out, int32_t planes, int64_t hw, const float* c_mean, const float* c_var,
                                   const float* s_mean, const float* s_var, float eps, void* stream) {
    if (!x || !out || !c_mean || !c_var || !s_mean || !s_var) return UAV_EINVAL;
    if (planes <= 0 || planes > 65535 || hw <= 0) return UAV_ESHAPE;
    const long long blocks = (hw + 1023) / 1024;
    if (blocks >= (1ll << 31)) return UAV_ESHAPE;
    hipLaunchKernelGGL(adain_apply_kernel, dim3((unsigned)blocks, planes), dim3(256), 0, (hipStream_t)stream, x, out, (long long)hw,
                       c_mean, c_var, s_mean, s_var, eps);
    return uav_launch_status();
}

extern "C" int uav_atrous_blur_f32(const float* src, float* low_out, float* high_inout, int32_t planes, int32_t h, int32_t w,
                                   int32_t radius, void* stream) {
    if (!src || !low_out || src == low_out) return UAV_EINVAL;
    if (planes <= 0 || planes > 65535 || h <= 0 || h > 65535 || w <= 0 || radius <= 0) return UAV_ESHAPE;
    hipLaunchKernelGGL(atrous_blur_kernel, dim3((w + 255) / 256, h, planes), dim3(256), 0, (hipStream_t)stream, src, low_out,
                       high_inout, h, w, radius);
    return uav_launch_status();
}

extern "C" int uav_resize_bicubic_f32(const float* src, float* dst, int32_t planes, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                                      float scale_h, float scale_w, void* stream) {
    if (!src || !dst) return UAV_EINVAL;
    if (planes <= 0 || planes > 65535 || hi <= 0 || wi <= 0 || ho <= 0 || ho > 65535 || wo <= 0 || scale_h <= 0.f || scale_w <= 0.f)
        return UAV_ESHAPE;
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3((wo + 255) / 256, ho, planes), dim3(256), 0, (hipStream_t)stream, src, dst, hi, wi,
                       ho, wo, scale_h, scale_w);
    return uav_launch_status();
}
