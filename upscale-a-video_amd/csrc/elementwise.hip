// HBM-bound / tiny kernels of the hot path for gfx950: layout edges, time-embedding MLP,
// fused CFG + DDIM step, flow-guided propagation step.  Each entry point cites the reference
// lines it replaces in include/uav_hip.h.
#include "uav_common.h"
#include <string.h>

namespace {

// ---------------------------------------------------------------------------------------------
// y[m][n] = post( sum_k pre(x[m][k]) * w[n][k] + b[n] ),  m <= 16.  One wave per output column.
template <int MMAX>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const half_t* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, int m,
                                                           int k, int n, int pre, int post) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= n) return;
    float acc[MMAX];
#pragma unroll
    for (int i = 0; i < MMAX; ++i) acc[i] = 0.f;
    for (int k0 = lane * 8; k0 < k; k0 += 64 * 8) {
        half8_t wv = *(const half8_t*)(w + (long long)col * k + k0);
#pragma unroll
        for (int i = 0; i < MMAX; ++i) {
            if (i < m) {
                const float4_t xa = *(const float4_t*)(x + (long long)i * k + k0);
                const float4_t xb = *(const float4_t*)(x + (long long)i * k + k0 + 4);
                float xs[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv = pre ? uav_silu(xs[e]) : xs[e];
                    acc[i] += xv * (float)wv[e];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MMAX; ++i) {
        if (i < m) {
            float s = wave_sum(acc[i]);
            if (lane == 0) {
                s += b ? b[col] : 0.f;
                y[(long long)i * n + col] = post ? uav_silu(s) : s;
            }
        }
    }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int m, int dim, int flip, float shift,
                                          float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= m * half) return;
    const int r = i / half, j = i % half;
    const float expo = -logf(10000.0f) * (float)j / ((float)half - shift);
    const float arg = t[r] * expf(expo);
    const float sn = sinf(arg), cs = cosf(arg);
    // diffusers get_timestep_embedding: cat[sin, cos]; flip_sin_to_cos swaps the halves
    out[(long long)r * dim + (flip ? half : 0) + j] = sn;
    out[(long long)r * dim + (flip ? 0 : half) + j] = cs;
}

// ---------------------------------------------------------------------------------------------
template <typename SrcT>
__global__ __launch_bounds__(256) void pack_nhwc_kernel(const SrcT* __restrict__ s1, int c1, const SrcT* __restrict__ s2,
                                                        int c2, half_t* __restrict__ dst, int c_pad, int n_batch,
                                                        int t_len, long long hw, float scale) {
    const long long rows = (long long)n_batch * t_len * hw;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const long long p = r % hw; const long long bt = r / hw;
    const int t = (int)(bt % t_len), b = (int)(bt / t_len);
    for (int cv = 0; cv < c_pad; cv += 8) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cv + e;
            float v = 0.f;
            if (c < c1) v = (float)s1[(((long long)b * c1 + c) * t_len + t) * hw + p];
            else if (c < c1 + c2) v = (float)s2[(((long long)b * c2 + (c - c1)) * t_len + t) * hw + p];
            o[e] = (half_t)(v * scale);
        }
        *(half8_t*)(dst + r * c_pad + cv) = o;
    }
}

template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void unpack_ncthw_kernel(const SrcT* __restrict__ src, int src_stride,
                                                           DstT* __restrict__ dst, int c, int n_batch, int t_len,
                                                           long long hw, float lo, float hi) {
    const long long rows = (long long)n_batch * t_len * hw;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const long long p = r % hw; const long long bt = r / hw;
    const int t = (int)(bt % t_len), b = (int)(bt / t_len);
    for (int ch = 0; ch < c; ++ch) {
        float v = (float)src[r * src_stride + ch];
        v = fminf(fmaxf(v, lo), hi);
        dst[(((long long)b * c + ch) * t_len + t) * hw + p] = (DstT)v;
    }
}

// ---------------------------------------------------------------------------------------------
// T = half_t: the reference's `.half()` arithmetic (the guided output is ROUNDED to fp16 before x0 is formed, as two
// separate fp16 tensor ops would); T = float: fp32 latents / model outputs (UNet stream_dtype = float32), nothing is rounded.
template <typename T>
__global__ __launch_bounds__(256) void cfg_ddim_v0_kernel(const T* __restrict__ eu, const T* __restrict__ ec,
                                                          const T* __restrict__ x, T* __restrict__ g_out,
                                                          T* __restrict__ x0_out, long long n, float guidance,
                                                          float ca, float cb, int clip, float range) {
    const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i0 >= n) return;
    const long long i1 = i0 + 8 <= n ? i0 + 8 : n;
    if (sizeof(T) == 2 && i0 + 8 <= n) {
        typedef half8_t V;
        const V u = *(const V*)(eu + i0), xv = *(const V*)(x + i0);
        V c = u, g, x0;
        if (ec) c = *(const V*)(ec + i0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gu = (float)u[e];
            const float gg = ec ? gu + guidance * ((float)c[e] - gu) : gu;
            g[e] = (half_t)gg;
            float v = ca * (float)xv[e] + cb * (float)g[e];
            if (clip) v = fminf(fmaxf(v, -range), range);
            x0[e] = (half_t)v;
        }
        *(V*)(g_out + i0) = g; *(V*)(x0_out + i0) = x0;
        return;
    }
    for (long long j = i0; j < i1; ++j) {
        const float gu = (float)eu[j];
        const float gg = ec ? gu + guidance * ((float)ec[j] - gu) : gu;
        const T gh = (T)gg;
        float v = ca * (float)x[j] + cb * (float)gh;
        if (clip) v = fminf(fmaxf(v, -range), range);
        g_out[j] = gh; x0_out[j] = (T)v;
    }
}

// prev = cx0*x0 + cdir*(em*g + es*sample + e0*x0)
template <typename T>
__global__ __launch_bounds__(256) void ddim_vt_kernel(const T* __restrict__ x0, const T* __restrict__ g,
                                                      const T* __restrict__ x, T* __restrict__ prev,
                                                      long long n, float cx0, float cdir, float em, float es, float e0,
                                                      int clip, float range) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = (float)x0[i];
    if (clip) a = fminf(fmaxf(a, -range), range);
    const float eps = em * (float)g[i] + es * (float)x[i] + e0 * a;
    prev[i] = (T)(cx0 * a + cdir * eps);
}

__global__ __launch_bounds__(256) void axpby_kernel(const half_t* __restrict__ x, const half_t* __restrict__ z,
                                                    half_t* __restrict__ y, long long n, float a, float b) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = (half_t)(a * (float)x[i] + b * (float)z[i]);
}

// fp32 rows -> fp16 rows (the VAE decoder's fp32 residual stream feeding a conv operand directly, i.e. the
// nearest-2x upsampler conv), 8 elements per thread.
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4_t a = *(const float4_t*)(x + i), b = *(const float4_t*)(x + i + 4);
        half8_t o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
        *(half8_t*)(y + i) = o;
    } else {
        for (long long j = i; j < n; ++j) y[j] = (half_t)x[j];
    }
}

// fp32 rows [M][C] -> fp16 rows [M][2C] = [hi | lo], hi = fp16(x), lo = fp16(x - hi): the stream as TWO MFMA operands (K
// doubled, weights repeated) where a conv reads it directly (down / up samplers); 8 elements per thread, C % 8 == 0.
__global__ __launch_bounds__(256) void cast_f32_hilo_kernel(const float* __restrict__ x, half_t* __restrict__ y, long long rows, int c) {
    const int per_row = c / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * per_row) return;
    const long long r = i / per_row;
    const int k = (int)(i % per_row) * 8;
    const float* src = x + r * c + k;
    const float4_t a = *(const float4_t*)src, b = *(const float4_t*)(src + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    half8_t hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (half_t)v[j]; lo[j] = (half_t)(v[j] - (float)hi[j]); }
    *(half8_t*)(y + r * 2 * c + k) = hi;
    *(half8_t*)(y + r * 2 * c + c + k) = lo;
}

// SFT fusion of the video VAE (Fuse_sft_block, resnet.py:76-78): out = dec + w*(dec*scale + shift) = dec*(1 + w*scale) + w*shift
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void sft_fuse_kernel(const TI* __restrict__ dec, const TI* __restrict__ scale,
                                                       const TI* __restrict__ shift, TO* __restrict__ out, long long n, float w) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = (float)dec[i];
    out[i] = (TO)(d + w * (d * (float)scale[i] + (float)shift[i]));
}

// ---------------------------------------------------------------------------------------------
// Propagation step.  R = rounding policy: fp16 replay of the reference's GPU arithmetic
// (flow_warp builds the grid in the latent dtype, propagation_module.py:123-132; ATen's
// grid_sampler_2d evaluates index/weights in scalar_t) or plain fp32 (matches the fp32 oracle).
struct RoundF16 { static UAV_DEVINL float r(float v) { return (float)(half_t)v; } };
struct RoundF32 { static UAV_DEVINL float r(float v) { return v; } };

template <typename R>
UAV_DEVINL void warp_coords(int x, int y, float fx, float fy, int w, int h, float& ix, float& iy) {
    const float vx = R::r((float)x + fx), vy = R::r((float)y + fy);
    const float dw = (float)(w - 1 > 1 ? w - 1 : 1), dh = (float)(h - 1 > 1 ? h - 1 : 1);
    const float gx = R::r(R::r(R::r(2.0f * vx) / dw) - 1.0f);
    const float gy = R::r(R::r(R::r(2.0f * vy) / dh) - 1.0f);
    // grid_sampler_unnormalize(align_corners=True): float arithmetic, one rounding to scalar_t
    ix = R::r(((gx + 1.f) / 2.f) * (float)(w - 1));
    iy = R::r(((gy + 1.f) / 2.f) * (float)(h - 1));
}

template <typename R, typename T>
UAV_DEVINL float sample_bilinear(const T* __restrict__ plane, int w, int h, float ix, float iy) {
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = R::r(R::r((float)x1 - ix) * R::r((float)y1 - iy));
    const float ne = R::r(R::r(ix - (float)x0) * R::r((float)y1 - iy));
    const float sw = R::r(R::r((float)x1 - ix) * R::r(iy - (float)y0));
    const float se = R::r(R::r(ix - (float)x0) * R::r(iy - (float)y0));
    float acc = 0.f;
    if (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h) acc = R::r(acc + R::r((float)plane[(long long)y0 * w + x0] * nw));
    if (x1 >= 0 && x1 < w && y0 >= 0 && y0 < h) acc = R::r(acc + R::r((float)plane[(long long)y0 * w + x1] * ne));
    if (x0 >= 0 && x0 < w && y1 >= 0 && y1 < h) acc = R::r(acc + R::r((float)plane[(long long)y1 * w + x0] * sw));
    if (x1 >= 0 && x1 < w && y1 >= 0 && y1 < h) acc = R::r(acc + R::r((float)plane[(long long)y1 * w + x1] * se));
    return acc;
}

template <typename R, typename T>
__global__ __launch_bounds__(256) void propagate_step_kernel(const T* __restrict__ prev, const T* __restrict__ cur,
                                                             const T* __restrict__ fprop, const T* __restrict__ fchk,
                                                             T* __restrict__ out, int c, int h, int w, long long fcs,
                                                             long long wcs, int nearest, float fuse, float a1, float a2) {
    // fcs / wcs: channel strides (elements) of the feature planes / flow planes
    const long long hw = (long long)h * w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    const int y = (int)(i / w), x = (int)(i % w);
    const float fx = (float)fprop[i], fy = (float)fprop[wcs + i];
    float ix, iy;
    warp_coords<R>(x, y, fx, fy, w, h, ix, iy);
    // forward-backward consistency (fbConsistencyCheck)
    const float bx = sample_bilinear<R, T>(fchk, w, h, ix, iy);
    const float by = sample_bilinear<R, T>(fchk + wcs, w, h, ix, iy);
    const float dx = R::r(fx + bx), dy = R::r(fy + by);
    const float ldiff = R::r(R::r(dx * dx) + R::r(dy * dy));
    const float lf = R::r(R::r(fx * fx) + R::r(fy * fy));
    const float lb = R::r(R::r(bx * bx) + R::r(by * by));
    const float mag = R::r(lf + lb);
    const float thr = R::r(R::r(a1 * mag) + a2);
    const bool valid = ldiff < thr;
    int xn = 0, yn = 0; bool inb = false;
    if (nearest) {
        xn = (int)nearbyintf(ix); yn = (int)nearbyintf(iy);
        inb = xn >= 0 && xn < w && yn >= 0 && yn < h;
    }
    for (int ch = 0; ch < c; ++ch) {
        const float cv = (float)cur[ch * fcs + i];
        float o = cv;
        if (valid) {
            float wv;
            if (nearest) wv = inb ? (float)prev[ch * fcs + (long long)yn * w + xn] : 0.f;
            else wv = sample_bilinear<R, T>(prev + ch * fcs, w, h, ix, iy);
            o = R::r(R::r(wv * fuse) + R::r(cv * (1.0f - fuse)));
        }
        out[ch * fcs + i] = (T)o;
    }
}

inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int uav_version(void) { return UAV_ABI_VERSION; }
#ifdef UAV_DEV_KERNELS
extern "C" int uav_has_dev_kernels(void) { return 1; }
#else
extern "C" int uav_has_dev_kernels(void) { return 0; }
#endif

extern "C" int uav_device_check(int dev, char* name_out) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return UAV_EINVAL;
    if (name_out) { strncpy(name_out, prop.gcnArchName, 63); name_out[63] = 0; }
    return strstr(prop.gcnArchName, "gfx950") ? 0 : UAV_EINVAL;
}

extern "C" int uav_linear_small(const float* x, const void* w, const float* b, float* y, int32_t m, int32_t k, int32_t n,
                                int32_t pre_act, int32_t post_act, void* stream) {
    if (!x || !w || !y) return UAV_EINVAL;
    if (m <= 0 || m > 16 || k <= 0 || (k % 8) || n <= 0) return UAV_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (m <= 2)
        hipLaunchKernelGGL(linear_small_kernel<2>, dim3((n + 3) / 4), dim3(256), 0, s, x, (const half_t*)w, b, y, m, k, n, pre_act, post_act);
    else
        hipLaunchKernelGGL(linear_small_kernel<16>, dim3((n + 3) / 4), dim3(256), 0, s, x, (const half_t*)w, b, y, m, k, n, pre_act, post_act);
    return uav_launch_status();
}

extern "C" int uav_timestep_embedding(const float* t, int32_t m, int32_t dim, int32_t flip_sin_to_cos, float freq_shift,
                                      float* out, void* stream) {
    if (!t || !out) return UAV_EINVAL;
    if (m <= 0 || dim <= 0 || (dim % 2)) return UAV_ESHAPE;
    const int total = m * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, m, dim,
                       flip_sin_to_cos, freq_shift, out);
    return uav_launch_status();
}

extern "C" int uav_pack_nhwc(const void* src1, int32_t c1, const void* src2, int32_t c2, int32_t src_is_f32, void* dst,
                             int32_t c_pad, int32_t n_batch, int32_t t_len, int64_t hw, float scale, void* stream) {
    if (!src1 || !dst) return UAV_EINVAL;
    if (c1 <= 0 || c2 < 0 || (c2 > 0 && !src2) || c_pad < c1 + c2 || (c_pad % 8) || n_batch <= 0 || t_len <= 0 || hw <= 0)
        return UAV_ESHAPE;
    const long long rows = (long long)n_batch * t_len * hw;
    hipStream_t s = (hipStream_t)stream;
    if (src_is_f32)
        hipLaunchKernelGGL(pack_nhwc_kernel<float>, dim3(nblk(rows, 256)), dim3(256), 0, s, (const float*)src1, c1,
                           (const float*)src2, c2, (half_t*)dst, c_pad, n_batch, t_len, (long long)hw, scale);
    else
        hipLaunchKernelGGL(pack_nhwc_kernel<half_t>, dim3(nblk(rows, 256)), dim3(256), 0, s, (const half_t*)src1, c1,
                           (const half_t*)src2, c2, (half_t*)dst, c_pad, n_batch, t_len, (long long)hw, scale);
    return uav_launch_status();
}

extern "C" int uav_unpack_ncthw(const void* src, int32_t src_stride, int32_t src_is_f32, void* dst, int32_t dst_is_f32,
                                int32_t c, int32_t n_batch, int32_t t_len, int64_t hw, float clamp_lo, float clamp_hi,
                                void* stream) {
    if (!src || !dst) return UAV_EINVAL;
    if (c <= 0 || src_stride < c || n_batch <= 0 || t_len <= 0 || hw <= 0) return UAV_ESHAPE;
    const long long rows = (long long)n_batch * t_len * hw;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(nblk(rows, 256)), b(256);
#define UNPACK(ST, DT) hipLaunchKernelGGL((unpack_ncthw_kernel<ST, DT>), g, b, 0, s, (const ST*)src, src_stride, (DT*)dst, c, \
                                          n_batch, t_len, (long long)hw, clamp_lo, clamp_hi)
    if (src_is_f32) { if (dst_is_f32) UNPACK(float, float); else UNPACK(float, half_t); }
    else            { if (dst_is_f32) UNPACK(half_t, float); else UNPACK(half_t, half_t); }
#undef UNPACK
    return uav_launch_status();
}

extern "C" int uav_cfg_ddim_v0(const void* eps_uncond, const void* eps_text, const void* sample, void* guided_out,
                               void* x0_out, int64_t n, float guidance, float coef_sample, float coef_eps, int32_t clip,
                               float clip_range, void* stream) {
    if (!eps_uncond || !sample || !guided_out || !x0_out || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(cfg_ddim_v0_kernel<half_t>, dim3(nblk(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)eps_uncond, (const half_t*)eps_text, (const half_t*)sample, (half_t*)guided_out,
                       (half_t*)x0_out, (long long)n, guidance, coef_sample, coef_eps, clip, clip_range);
    return uav_launch_status();
}

extern "C" int uav_cfg_ddim_v0_f32(const float* eps_uncond, const float* eps_text, const float* sample, float* guided_out,
                                   float* x0_out, int64_t n, float guidance, float coef_sample, float coef_eps, int32_t clip,
                                   float clip_range, void* stream) {
    if (!eps_uncond || !sample || !guided_out || !x0_out || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(cfg_ddim_v0_kernel<float>, dim3(nblk(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, eps_uncond,
                       eps_text, sample, guided_out, x0_out, (long long)n, guidance, coef_sample, coef_eps, clip, clip_range);
    return uav_launch_status();
}

extern "C" int uav_ddim_vt(const void* x0, const void* guided, const void* sample, void* prev_out, int64_t n,
                           float coef_x0, float coef_dir, float eps_from_model, float eps_from_sample, float eps_from_x0,
                           int32_t clip, float clip_range, void* stream) {
    if (!x0 || !guided || !sample || !prev_out || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(ddim_vt_kernel<half_t>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x0,
                       (const half_t*)guided, (const half_t*)sample, (half_t*)prev_out, (long long)n, coef_x0, coef_dir,
                       eps_from_model, eps_from_sample, eps_from_x0, clip, clip_range);
    return uav_launch_status();
}

extern "C" int uav_ddim_vt_f32(const float* x0, const float* guided, const float* sample, float* prev_out, int64_t n,
                               float coef_x0, float coef_dir, float eps_from_model, float eps_from_sample, float eps_from_x0,
                               int32_t clip, float clip_range, void* stream) {
    if (!x0 || !guided || !sample || !prev_out || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(ddim_vt_kernel<float>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x0, guided, sample,
                       prev_out, (long long)n, coef_x0, coef_dir, eps_from_model, eps_from_sample, eps_from_x0, clip, clip_range);
    return uav_launch_status();
}

extern "C" int uav_axpby_f16(const void* x, const void* z, void* y, int64_t n, float a, float b, void* stream) {
    if (!x || !z || !y || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (const half_t*)z, (half_t*)y, (long long)n, a, b);
    return uav_launch_status();
}

extern "C" int uav_cast_f32_f16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return UAV_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return UAV_EALIGN;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(nblk((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (half_t*)y,
                       (long long)n);
    return uav_launch_status();
}

extern "C" int uav_cast_f32_hilo(const float* x, void* y, int64_t rows, int32_t c, void* stream) {
    if (!x || !y || rows <= 0 || c <= 0) return UAV_EINVAL;
    if (c % 8) return UAV_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return UAV_EALIGN;
    hipLaunchKernelGGL(cast_f32_hilo_kernel, dim3(nblk(rows * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (half_t*)y,
                       (long long)rows, c);
    return uav_launch_status();
}

extern "C" int uav_sft_fuse(const void* dec, const void* scale, const void* shift, void* out, int64_t n, float w,
                            int32_t in_f32, int32_t out_f32, void* stream) {
    if (!dec || !scale || !shift || !out || n <= 0) return UAV_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblk(n, 256)), blk(256);
    if (in_f32 && out_f32)
        hipLaunchKernelGGL((sft_fuse_kernel<float, float>), grid, blk, 0, s, (const float*)dec, (const float*)scale, (const float*)shift, (float*)out, (long long)n, w);
    else if (in_f32)
        hipLaunchKernelGGL((sft_fuse_kernel<float, half_t>), grid, blk, 0, s, (const float*)dec, (const float*)scale, (const float*)shift, (half_t*)out, (long long)n, w);
    else if (out_f32)
        hipLaunchKernelGGL((sft_fuse_kernel<half_t, float>), grid, blk, 0, s, (const half_t*)dec, (const half_t*)scale, (const half_t*)shift, (float*)out, (long long)n, w);
    else
        hipLaunchKernelGGL((sft_fuse_kernel<half_t, half_t>), grid, blk, 0, s, (const half_t*)dec, (const half_t*)scale, (const half_t*)shift, (half_t*)out, (long long)n, w);
    return uav_launch_status();
}

extern "C" int uav_propagate_step_f16(const void* feat_prev, const void* feat_cur, const void* flow_prop,
                                      const void* flow_check, void* out, int32_t c, int32_t h, int32_t w,
                                      int64_t feat_chan_stride, int64_t flow_chan_stride, int32_t nearest,
                                      int32_t coord_f16, float fuse_scale, float alpha1, float alpha2, void* stream) {
    if (!feat_prev || !feat_cur || !flow_prop || !flow_check || !out) return UAV_EINVAL;
    if (c <= 0 || h <= 0 || w <= 0) return UAV_ESHAPE;
    const long long hw = (long long)h * w;
    if (feat_chan_stride < hw || flow_chan_stride < hw) return UAV_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (coord_f16)
        hipLaunchKernelGGL((propagate_step_kernel<RoundF16, half_t>), dim3(nblk(hw, 256)), dim3(256), 0, s, (const half_t*)feat_prev,
                           (const half_t*)feat_cur, (const half_t*)flow_prop, (const half_t*)flow_check, (half_t*)out, c, h,
                           w, (long long)feat_chan_stride, (long long)flow_chan_stride, nearest, fuse_scale, alpha1, alpha2);
    else
        hipLaunchKernelGGL((propagate_step_kernel<RoundF32, half_t>), dim3(nblk(hw, 256)), dim3(256), 0, s, (const half_t*)feat_prev,
                           (const half_t*)feat_cur, (const half_t*)flow_prop, (const half_t*)flow_check, (half_t*)out, c, h,
                           w, (long long)feat_chan_stride, (long long)flow_chan_stride, nearest, fuse_scale, alpha1, alpha2);
    return uav_launch_status();
}

// fp32 planes, fp32 flows, fp32 coordinates: the arithmetic of the reference's fp32 run (pipeline:651 casts the flows to the
// latent dtype, so an fp32 pipeline warps fp32 values on fp32 grids).
extern "C" int uav_propagate_step_f32(const float* feat_prev, const float* feat_cur, const float* flow_prop,
                                      const float* flow_check, float* out, int32_t c, int32_t h, int32_t w,
                                      int64_t feat_chan_stride, int64_t flow_chan_stride, int32_t nearest, float fuse_scale,
                                      float alpha1, float alpha2, void* stream) {
    if (!feat_prev || !feat_cur || !flow_prop || !flow_check || !out) return UAV_EINVAL;
    if (c <= 0 || h <= 0 || w <= 0) return UAV_ESHAPE;
    const long long hw = (long long)h * w;
    if (feat_chan_stride < hw || flow_chan_stride < hw) return UAV_ESHAPE;
    hipLaunchKernelGGL((propagate_step_kernel<RoundF32, float>), dim3(nblk(hw, 256)), dim3(256), 0, (hipStream_t)stream,
                       feat_prev, feat_cur, flow_prop, flow_check, out, c, h, w, (long long)feat_chan_stride,
                       (long long)flow_chan_stride, nearest, fuse_scale, alpha1, alpha2);
    return uav_launch_status();
}
