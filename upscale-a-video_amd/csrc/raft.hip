// K11b — non-GEMM kernels of the RAFT optical-flow network (fp32, channels-last rows), replacing
// the ATen ops of models_video/RAFT/{extractor,corr,update,raft}.py:
//   instance norm (+ReLU)            extractor.py:27-30,48-49 (nn.InstanceNorm2d, no affine)
//   correlation pyramid pooling      corr.py:23-27 (avg_pool2d 2x2)
//   9x9x4 correlation lookup         corr.py:29-50 + utils/utils.py:57-71 (bilinear grid_sample,
//                                    align_corners=True, zero padding), written as 324(+pad) channels
//   ConvGRU gate arithmetic          update.py:44-58
//   convex 8x upsampling             raft.py:73-85 (softmax over 9 + 3x3 unfold)
//   small row utilities              column copies for the channel concats, axpby, add+ReLU
// All are HBM/latency bound and tiny next to the encoders' and GRU's convolutions.
#include "uav_common.h"

namespace {

// ---- instance norm: one workgroup per (image, 32-channel slab); two-pass mean / variance ------------
__global__ __launch_bounds__(256) void instnorm_kernel(const float* __restrict__ x, float* __restrict__ y, int hw, int c,
                                                       float eps, int relu) {
    __shared__ float red[8][33];
    __shared__ float s_mean[32], s_rstd[32];
    const int img = blockIdx.y, c0 = blockIdx.x * 32;
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;                 // 32 channels x 8 row lanes
    const int ch = c0 + cl;
    const bool ok = ch < c;
    const float* base = x + (long long)img * hw * c;
    float s = 0.f;
    if (ok) for (int r = rl; r < hw; r += 8) s += base[(long long)r * c + ch];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0) { float a = 0.f; for (int k = 0; k < 8; ++k) a += red[k][cl]; s_mean[cl] = a / hw; }
    __syncthreads();
    const float mean = s_mean[cl];
    float q = 0.f;
    if (ok) for (int r = rl; r < hw; r += 8) { float d = base[(long long)r * c + ch] - mean; q += d * d; }
    red[rl][cl] = q;
    __syncthreads();
    if (rl == 0) { float a = 0.f; for (int k = 0; k < 8; ++k) a += red[k][cl]; s_rstd[cl] = rsqrtf(a / hw + eps); }
    __syncthreads();
    const float rstd = s_rstd[cl];
    float* ob = y + (long long)img * hw * c;
    if (ok) for (int r = rl; r < hw; r += 8) {
        float v = (base[(long long)r * c + ch] - mean) * rstd;
        ob[(long long)r * c + ch] = relu ? fmaxf(v, 0.f) : v;
    }
}

// ---- elementwise -------------------------------------------------------------------------------------
__global__ void add_relu_kernel(const float* a, const float* b, float* o, long long n, int relu) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = a[i] + b[i]; o[i] = relu ? fmaxf(v, 0.f) : v; }
}
__global__ void axpby_f32_kernel(const float* x, const float* z, float* y, long long n, float a, float b) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + b * z[i];
}
// dst[r][dcol + j] = act(src[r][scol + j]) for j < ncols
__global__ void copy_cols_kernel(const float* src, int sstride, int scol, float* dst, int dstride, int dcol, int ncols,
                                 long long rows, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ncols) return;
    const long long r = i / ncols; const int j = (int)(i - r * ncols);
    float v = src[r * sstride + scol + j];
    if (act == 1) v = fmaxf(v, 0.f); else if (act == 4) v = tanhf(v);
    dst[r * dstride + dcol + j] = v;
}
// rh = r * h with r = zr[:, c + j] (zr rows hold z | r), h rows of c channels
__global__ void gru_rh_kernel(const float* zr, const float* h, float* rh, long long rows, int c) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c) return;
    const long long r = i / c; const int j = (int)(i - r * c);
    rh[i] = zr[r * 2 * c + c + j] * h[i];
}
// h = (1 - z) * h + z * q
__global__ void gru_blend_kernel(const float* zr, const float* q, float* h, long long rows, int c) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c) return;
    const long long r = i / c; const int j = (int)(i - r * c);
    const float z = zr[r * 2 * c + j];
    h[i] = (1.0f - z) * h[i] + z * q[i];
}

// ---- correlation pyramid ------------------------------------------------------------------------------
// src: [P][h][w] (row stride `sstride` floats per P), dst: [P][h/2][w/2] contiguous
__global__ void avgpool2_kernel(const float* src, long long sstride, int h, int w, float* dst, long long p_count) {
    const int h2 = h / 2, w2 = w / 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p_count * h2 * w2) return;
    const long long pp = i / (h2 * w2); const int rem = (int)(i - pp * h2 * w2);
    const int y = rem / w2, x = rem - y * w2;
    const float* s = src + pp * sstride + (long long)(2 * y) * w + 2 * x;
    dst[i] = 0.25f * (s[0] + s[1] + s[w] + s[w + 1]);
}

// ---- bilinear plane resize (align_corners = False) ---------------------------------------------------
// RAFT_bi pre-resizes frames to multiples of 8 (`F.interpolate(..., mode='trilinear')` with T unchanged, i.e.
// bilinear per frame, raft_bi.py:53) and resizes the flows back (`resize_flow_pytorch`, :11-16).  The reference
// scales `flow[:, :, 0]` and `flow[:, :, 1]` afterwards, which indexes ROWS 0 and 1 of both channels, not the
// channels: row0_scale / row1_scale reproduce exactly that.
__global__ void resize_bilinear_kernel(const float* src, float* dst, long long planes, int hi, int wi, int ho, int wo,
                                       float sy, float sx, float row0_scale, float row1_scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * ho * wo) return;
    const long long pl = i / ((long long)ho * wo); const int rem = (int)(i - pl * ho * wo);
    const int y = rem / wo, x = rem - y * wo;
    const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.0f);
    const int y0 = min((int)fy, hi - 1), x0 = min((int)fx, wi - 1);
    const int y1 = y0 + (y0 < hi - 1), x1 = x0 + (x0 < wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* s = src + pl * (long long)hi * wi;
    const float top = (1.0f - lx) * s[(long long)y0 * wi + x0] + lx * s[(long long)y0 * wi + x1];
    const float bot = (1.0f - lx) * s[(long long)y1 * wi + x0] + lx * s[(long long)y1 * wi + x1];
    float v = (1.0f - ly) * top + ly * bot;
    if (y == 0) v *= row0_scale;
    if (y == 1) v *= row1_scale;
    dst[i] = v;
}

struct LookupArgs {
    const float* lvl[4]; long long stride[4]; int h[4], w[4];
    const float* coords; int coord_stride;       // rows [P][coord_stride]: x, y
    float* out; int out_stride; long long p_count; int radius;
};

UAV_DEVINL float bilin(const float* img, int h, int w, float x, float y) {
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = x - x0f, ay = y - y0f;
    float v = 0.f;
    if (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h) v += img[y0 * w + x0] * (1.f - ax) * (1.f - ay);
    if (x1 >= 0 && x1 < w && y0 >= 0 && y0 < h) v += img[y0 * w + x1] * ax * (1.f - ay);
    if (x0 >= 0 && x0 < w && y1 >= 0 && y1 < h) v += img[y1 * w + x0] * (1.f - ax) * ay;
    if (x1 >= 0 && x1 < w && y1 >= 0 && y1 < h) v += img[y1 * w + x1] * ax * ay;
    return v;
}

// one thread per (pixel, level, window entry): out[p][lvl*81 + a*9 + b] = sample(lvl, x/2^l + (a-r), y/2^l + (b-r))
// (the reference adds the FIRST meshgrid component, built from `dy`, to x: corr.py:37-43)
__global__ __launch_bounds__(256) void corr_lookup_kernel(LookupArgs a) {
    const int win = 2 * a.radius + 1, per = win * win, tot = 4 * per;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.p_count * tot) return;
    const long long pp = i / tot; const int k = (int)(i - pp * tot);
    const int l = k / per, e = k - l * per, ea = e / win, eb = e - ea * win;
    const float inv = 1.0f / (float)(1 << l);
    const float x = a.coords[pp * a.coord_stride] * inv + (float)(ea - a.radius);
    const float y = a.coords[pp * a.coord_stride + 1] * inv + (float)(eb - a.radius);
    a.out[pp * a.out_stride + k] = bilin(a.lvl[l] + pp * a.stride[l], a.h[l], a.w[l], x, y);
}

// ---- convex upsampling: flow rows [N*h*w][fs] (x,y), mask rows [N*h*w][576] = (9, 8, 8) -> (N,2,8h,8w) planar --
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow, int fs, const float* __restrict__ mask,
                                                              float* __restrict__ out, int n, int h, int w) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per output pixel
    const long long total = (long long)n * 64 * h * w;
    if (i >= total) return;
    const int W8 = 8 * w, H8 = 8 * h;
    const int X = (int)(i % W8); const long long t = i / W8; const int Y = (int)(t % H8); const int b = (int)(t / H8);
    const int x = X >> 3, sx = X & 7, y = Y >> 3, sy = Y & 7;
    const float* m = mask + ((long long)(b * h + y) * w + x) * 576 + sy * 8 + sx;       // [k][sy][sx], k stride 64
    float mv[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { mv[k] = m[k * 64]; mx = fmaxf(mx, mv[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { mv[k] = expf(mv[k] - mx); den += mv[k]; }
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;                       // F.unfold 3x3, padding 1
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const float* f = flow + ((long long)(b * h + yy) * w + xx) * fs;
            ox += mv[k] * 8.0f * f[0]; oy += mv[k] * 8.0f * f[1];
        }
    }
    const long long plane = (long long)H8 * W8;
    out[(long long)b * 2 * plane + (long long)Y * W8 + X] = ox / den;
    out[(long long)b * 2 * plane + plane + (long long)Y * W8 + X] = oy / den;
}

inline unsigned nb(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int uav_instnorm_f32(const float* x, float* y, int32_t n_img, int32_t hw, int32_t c, float eps, int32_t relu,
                                void* stream) {
    if (!x || !y || n_img <= 0 || hw <= 0 || c <= 0 || n_img > 65535) return UAV_EINVAL;
    hipLaunchKernelGGL(instnorm_kernel, dim3((c + 31) / 32, n_img), dim3(256), 0, (hipStream_t)stream, x, y, hw, c, eps, relu);
    return uav_launch_status();
}
extern "C" int uav_add_relu_f32(const float* a, const float* b, float* out, int64_t n, int32_t relu, void* stream) {
    if (!a || !b || !out || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(add_relu_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n, relu);
    return uav_launch_status();
}
extern "C" int uav_axpby_f32(const float* x, const float* z, float* y, int64_t n, float a, float b, void* stream) {
    if (!x || !z || !y || n <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(axpby_f32_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, x, z, y, (long long)n, a, b);
    return uav_launch_status();
}
extern "C" int uav_copy_cols_f32(const float* src, int32_t src_stride, int32_t src_col, float* dst, int32_t dst_stride,
                                 int32_t dst_col, int32_t ncols, int64_t rows, int32_t act, void* stream) {
    if (!src || !dst || ncols <= 0 || rows <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(nb(rows * ncols)), dim3(256), 0, (hipStream_t)stream, src, src_stride, src_col,
                       dst, dst_stride, dst_col, ncols, (long long)rows, act);
    return uav_launch_status();
}
extern "C" int uav_gru_gates_f32(const float* zr, const float* h_in, const float* q, float* out, int64_t rows, int32_t c,
                                 int32_t mode, void* stream) {
    // mode 0: out = r*h (q unused) ; mode 1: out(=h, in place allowed) = (1-z)*h + z*q
    if (!zr || !h_in || !out || rows <= 0 || c <= 0) return UAV_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(gru_rh_kernel, dim3(nb(rows * c)), dim3(256), 0, s, zr, h_in, out, (long long)rows, c);
    else {
        if (!q || out != h_in) return UAV_EINVAL;
        hipLaunchKernelGGL(gru_blend_kernel, dim3(nb(rows * c)), dim3(256), 0, s, zr, q, out, (long long)rows, c);
    }
    return uav_launch_status();
}
extern "C" int uav_resize_bilinear_f32(const float* src, float* dst, int64_t planes, int32_t hi, int32_t wi, int32_t ho,
                                       int32_t wo, float row0_scale, float row1_scale, void* stream) {
    if (!src || !dst || planes <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(nb(planes * ho * wo)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       (long long)planes, hi, wi, ho, wo, (float)hi / (float)ho, (float)wi / (float)wo, row0_scale, row1_scale);
    return uav_launch_status();
}
extern "C" int uav_avgpool2_f32(const float* src, int64_t src_stride, int32_t h, int32_t w, float* dst, int64_t p_count,
                                void* stream) {
    if (!src || !dst || h < 2 || w < 2 || p_count <= 0) return UAV_EINVAL;
    hipLaunchKernelGGL(avgpool2_kernel, dim3(nb(p_count * (h / 2) * (w / 2))), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)src_stride, h, w, dst, (long long)p_count);
    return uav_launch_status();
}
extern "C" int uav_corr_lookup_f32(const float* const* levels, const int64_t* strides, const int32_t* hs, const int32_t* ws,
                                   const float* coords, int32_t coord_stride, float* out, int32_t out_stride,
                                   int64_t p_count, int32_t radius, void* stream) {
    if (!levels || !strides || !hs || !ws || !coords || !out || p_count <= 0 || radius <= 0) return UAV_EINVAL;
    LookupArgs a;
    for (int l = 0; l < 4; ++l) { a.lvl[l] = levels[l]; a.stride[l] = strides[l]; a.h[l] = hs[l]; a.w[l] = ws[l]; }
    a.coords = coords; a.coord_stride = coord_stride; a.out = out; a.out_stride = out_stride; a.p_count = p_count; a.radius = radius;
    const int win = 2 * radius + 1;
    if (out_stride < 4 * win * win) return UAV_ESHAPE;
    hipLaunchKernelGGL(corr_lookup_kernel, dim3(nb(p_count * 4 * win * win)), dim3(256), 0, (hipStream_t)stream, a);
    return uav_launch_status();
}
extern "C" int uav_convex_upsample_f32(const float* flow, int32_t flow_stride, const float* mask, float* out, int32_t n,
                                       int32_t h, int32_t w, void* stream) {
    if (!flow || !mask || !out || n <= 0 || h <= 0 || w <= 0 || flow_stride < 2) return UAV_EINVAL;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3(nb((long long)n * 64 * h * w)), dim3(256), 0, (hipStream_t)stream, flow,
                       flow_stride, mask, out, n, h, w);
    return uav_launch_status();
}
