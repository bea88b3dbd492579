// K3 — GroupNorm statistics / apply(+SiLU) and LayerNorm for channels-last fp16 rows (gfx950).
//
// GroupNorm in the reference is applied to 5-D (b,c,t,h,w) tensors — statistics over
// (C/G, T, H, W) per batch element (resnet.py:267,278; unet_video.py:567; vae_video.py:401) —
// or per frame on (b t) c h w (attention.py:374; unet_blocks.py:740).  Both are the same
// kernel here: an "instance" is a run of `rows_per_inst` consecutive channels-last rows
// (T*H*W rows, or H*W rows for the per-frame flavour).
//
// Pass 1 (HBM-bound, reads x once = 2 B/element): per-(instance, chunk) per-channel fp32
// (sum, sumsq) partials.  Pass 2 (tiny): fp64 reduction over chunks and the channels of each
// group -> per-(instance, channel) fp32 scale = gamma*rstd, shift = beta - mean*rstd*gamma.
// Pass 3 (HBM-bound, 2 B read + 2 B written per element): y = act(x*scale + shift).
// The input may be two channel-concatenated tensors (skip connections, unet_blocks.py:563):
// groups may straddle the seam (1536 channels / 32 groups = 48), which per-channel partials
// handle for free.  Deterministic: no atomics.
#include "uav_common.h"
#include <stdlib.h>

namespace {

constexpr int GN_MAX_CHUNKS = 2048;

struct GnSrc {
    const char* x1; const char* x2; int c1, c2;
    long long rows2;      // 0, or the row count of x2 when it is read batch-broadcast (row r >= rows2 reads row r - rows2)
};

// 8 consecutive channels (vector index v inside the concatenated row) as fp32.  F32 = the rows are fp32 (fp32
// residual stream of the VAE decoder: conv outputs are normalised without an intermediate fp16 rounding).
template <bool F32>
UAV_DEVINL void gn_load8(const GnSrc& s, long long row, int v, float (&f)[8]) {
    const int v1 = s.c1 >> 3;
    const bool first = v < v1;
    const char* base = first ? s.x1 : s.x2;
    const long long row2 = (s.rows2 && row >= s.rows2) ? row - s.rows2 : row;
    const long long e = first ? row * s.c1 + (long long)v * 8 : row2 * s.c2 + (long long)(v - v1) * 8;
    if (F32) {
        const float4_t a = *(const float4_t*)(base + e * 4), b = *(const float4_t*)(base + e * 4 + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
    } else {
        const half8_t x = *(const half8_t*)(base + e * 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)x[j];
    }
}

template <bool F32, int U>
__global__ __launch_bounds__(256) void gn_partial_kernel(GnSrc s, long long rows_per_inst, int chunks, int c_real, int groups,
                                                         float* __restrict__ ws) {
    __shared__ float red[256 * 16];
    const int c = s.c1 + s.c2, cvec = c >> 3;
    const int rpp = 256 / cvec;                       // rows per pass (>=1, cvec <= 256)
    const int tid = threadIdx.x;
    const int v = tid % cvec, ro = tid / cvec;
    const bool active = ro < rpp;
    const int inst = blockIdx.y, chunk = blockIdx.x;
    const long long rows_per_chunk = (rows_per_inst + chunks - 1) / chunks;
    const long long r0 = (long long)chunk * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk; if (r1 > rows_per_inst) r1 = rows_per_inst;
    float sm[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sm[j] = 0.f; sq[j] = 0.f; }
    if (active) {
        // U independent 16-B loads in flight per thread
        for (long long r = r0 + ro; r < r1; r += (long long)rpp * U) {
            float x[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long rr = r + (long long)u * rpp;
                if (rr < r1) gn_load8<F32>(s, inst * rows_per_inst + rr, v, x[u]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[u][j] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = x[u][j]; sm[j] += f; sq[j] += f * f; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = sm[j]; red[tid * 16 + 8 + j] = sq[j]; }
    __syncthreads();
    float a[16];
    if (tid < cvec) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a[j] = 0.f;
            for (int q = 0; q < rpp; ++q) a[j] += red[(q * cvec + tid) * 16 + j];
        }
    }
    __syncthreads();
    // per-channel totals of this chunk -> LDS [0:c] sums, [c:2c] sums of squares, then one (sum, sumsq) pair per GROUP:
    // the finalize pass reads chunks x 2 floats per group instead of chunks x 2 x (C/G) (it was a third of the statistics
    // time at 2048 chunks: 36 us per launch)
    if (tid < cvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 8 + j] = a[j]; red[c + tid * 8 + j] = a[8 + j]; }
    }
    __syncthreads();
    const int cpg = c_real / groups;
    float* dst = ws + (long long)(inst * chunks + chunk) * 2 * groups;
    for (int g = tid; g < groups; g += 256) {
        float gs = 0.f, gq = 0.f;
        for (int k = 0; k < cpg; ++k) { gs += red[g * cpg + k]; gq += red[c + g * cpg + k]; }
        dst[g] = gs; dst[groups + g] = gq;
    }
}

// One workgroup per (group, instance): fp64 reduction of the per-chunk (sum, sumsq) pairs of the group, then scale/shift
// for its channels.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ ws, int chunks, int c, int c_real,
                                                          int groups, long long rows_per_inst, float eps,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, inst = blockIdx.y, tid = threadIdx.x;
    const int cpg = c_real / groups;
    const float* base = ws + (long long)inst * chunks * 2 * groups;
    double a = 0.0, b = 0.0;
    for (int k = tid; k < chunks; k += 256) {
        a += base[(long long)k * 2 * groups + g];
        b += base[(long long)k * 2 * groups + groups + g];
    }
    rs[tid] = a; rq[tid] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    const double n = (double)rows_per_inst * cpg;
    const double mean = rs[0] / n;
    double var = rq[0] / n - mean * mean; if (var < 0.0) var = 0.0;
    const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = tid; j < cpg; j += 256) {
        const int ch = g * cpg + j;
        const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
        scale[(long long)inst * c + ch] = ga * fr;
        shift[(long long)inst * c + ch] = be - fm * fr * ga;
    }
    // padding channels (c_real <= ch < c) carry scale = shift = 0
    if (g == 0)
        for (int ch = c_real + tid; ch < c; ch += 256) { scale[(long long)inst * c + ch] = 0.f; shift[(long long)inst * c + ch] = 0.f; }
}

// Same result from the partials a conv epilogue wrote (conv_gemm.hip: conv_gn_store): [2][groups][chunks_total], the
// instance's chunks are contiguous, so the reads are coalesced; fp64 tree as above.
__global__ __launch_bounds__(1024) void gn_finalize_partials_kernel(const float* __restrict__ ws, long long chunks_total,
                                                                    long long chunks, int c, int groups, long long rows_per_inst,
                                                                    float eps, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ scale,
                                                                    float* __restrict__ shift) {
    __shared__ double rs[1024], rq[1024];
    const int g = blockIdx.x, inst = blockIdx.y, tid = threadIdx.x;
    const int cpg = c / groups;
    const float* ps = ws + (long long)g * chunks_total + (long long)inst * chunks;
    const float* pq = ps + (long long)groups * chunks_total;
    double a = 0.0, b = 0.0;
    for (long long k = tid; k < chunks; k += 1024) { a += ps[k]; b += pq[k]; }
    rs[tid] = a; rq[tid] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    const double n = (double)rows_per_inst * cpg;
    const double mean = rs[0] / n;
    double var = rq[0] / n - mean * mean; if (var < 0.0) var = 0.0;
    const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = tid; j < cpg; j += 1024) {
        const int ch = g * cpg + j;
        const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
        scale[(long long)inst * c + ch] = ga * fr;
        shift[(long long)inst * c + ch] = be - fm * fr * ga;
    }
}

// raw (optional): the UN-normalised input, concatenated and rounded to fp16 — the MFMA operand of the 1x1 shortcut conv
// of a block whose input is an fp32 stream (+ skip tensor); written here because this pass has the fp32 values in
// registers anyway (a separate cast pass would read them a second time).
// Two-source form (channel-concatenated input [x1 | x2] of an up-block ResNet, never materialised): each source carries the
// partials ITS producer wrote for its own group size; an output group of cpg channels lies inside one source (the host
// checks c1 % cpg == 0) and is the sum of cpg / cpgS consecutive producer groups.  A source that exists once for both batch
// entries (skip tensor of the CFG-shared head) serves instance i with the chunks of instance i % inst2.
struct GnSrcPartials { const float* ws; long long chunks_total, chunks; int c, groups, n_inst; };
__global__ __launch_bounds__(1024) void gn_finalize_partials2_kernel(GnSrcPartials s1, GnSrcPartials s2, int groups,
                                                                     long long rows_per_inst, float eps,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double rs[1024], rq[1024];
    const int g = blockIdx.x, inst = blockIdx.y, tid = threadIdx.x;
    const int c = s1.c + s2.c, cpg = c / groups, ch0 = g * cpg;
    const bool first = ch0 < s1.c;
    const GnSrcPartials& s = first ? s1 : s2;
    const int cpgs = s.c / s.groups, k = cpg / cpgs, pg0 = (first ? ch0 : ch0 - s1.c) / cpgs;
    const int si = inst % s.n_inst;
    double a = 0.0, b = 0.0;
    for (int j = 0; j < k; ++j) {
        const float* ps = s.ws + (long long)(pg0 + j) * s.chunks_total + (long long)si * s.chunks;
        const float* pq = ps + (long long)s.groups * s.chunks_total;
        for (long long q = tid; q < s.chunks; q += 1024) { a += ps[q]; b += pq[q]; }
    }
    rs[tid] = a; rq[tid] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    const double n = (double)rows_per_inst * cpg;
    const double mean = rs[0] / n;
    double var = rq[0] / n - mean * mean; if (var < 0.0) var = 0.0;
    const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = tid; j < cpg; j += 1024) {
        const int ch = ch0 + j;
        const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
        scale[(long long)inst * c + ch] = ga * fr;
        shift[(long long)inst * c + ch] = be - fm * fr * ga;
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnSrc s, long long rows_per_inst, int chunks,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int silu, char* __restrict__ y, char* __restrict__ raw, int raw_hilo) {
    const int c = s.c1 + s.c2, cvec = c >> 3;
    const int rpp = 256 / cvec;
    const int tid = threadIdx.x;
    const int v = tid % cvec, ro = tid / cvec;
    if (ro >= rpp) return;
    const int inst = blockIdx.y, chunk = blockIdx.x;
    const long long rows_per_chunk = (rows_per_inst + chunks - 1) / chunks;
    const long long r0 = (long long)chunk * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk; if (r1 > rows_per_inst) r1 = rows_per_inst;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[(long long)inst * c + v * 8 + j]; sh[j] = shift[(long long)inst * c + v * 8 + j]; }
    for (long long r = r0 + ro; r < r1; r += rpp) {
        const long long row = inst * rows_per_inst + r;
        float x[8];
        gn_load8<F32>(s, row, v, x);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = x[j] * sc[j] + sh[j];
            if (silu) f = uav_silu(f);
            o[j] = (half_t)f;
        }
        *(half8_t*)(y + (row * c + (long long)v * 8) * 2) = o;
        if (raw) {
            // raw_hilo: rows of 2c values [hi | lo], hi = fp16(x), lo = fp16(x - hi): x to ~22 bits as TWO fp16 operands (the
            // shortcut conv then runs over K = 2c with its weights repeated), so the stream is not rounded to fp16 on that path
            half8_t r, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) { r[j] = (half_t)x[j]; lo[j] = (half_t)(x[j] - (float)r[j]); }
            const long long rs = raw_hilo ? 2ll * c : (long long)c;
            *(half8_t*)(raw + (row * rs + (long long)v * 8) * 2) = r;
            if (raw_hilo) *(half8_t*)(raw + (row * rs + c + (long long)v * 8) * 2) = lo;
        }
    }
}

// LayerNorm: one wave per row, row held in registers (c <= 2048), two-pass mean/variance.  (Round 3 tried hoisting gamma / beta
// out of the row loop and prefetching the next row: 196 -> 228 ms per clip, dropped — the extra registers cost occupancy, and the
// in-order vmcnt puts the prefetch's wait in front of the store anyway.)  F32: the rows are the fp32
// residual stream (UNet stream_dtype = float32); the output is always the fp16 MFMA operand of the projection that follows.
template <int NV, bool F32>
__global__ __launch_bounds__(256) void layernorm_kernel(const char* __restrict__ x, char* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        long long rows, int c, float eps) {
    const int lane = threadIdx.x & 63;
    const int cvec = c >> 3;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long row = wave_id; row < rows; row += nwaves) {
        float xv[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + i * 64;
            if (v < cvec) {
                if (F32) {
                    const float4_t a = *(const float4_t*)(x + (row * c + (long long)v * 8) * 4);
                    const float4_t b = *(const float4_t*)(x + (row * c + (long long)v * 8) * 4 + 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xv[i][j] = a[j]; xv[i][4 + j] = b[j]; }
                } else {
                    const half8_t h = *(const half8_t*)(x + (row * c + (long long)v * 8) * 2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[i][j] = (float)h[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += xv[i][j];
            }
        }
        const float mean = wave_sum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + i * 64;
            if (v < cvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float d = xv[i][j] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + i * 64;
            if (v < cvec) {
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = v * 8 + j;
                    o[j] = (half_t)((xv[i][j] - mean) * rstd * gamma[ch] + beta[ch]);
                }
                *(half8_t*)(y + (row * c + (long long)v * 8) * 2) = o;
            }
        }
    }
}

int gn_variant() {
    static const int v = [] { const char* e = getenv("UAV_GN_VAR"); return e ? atoi(e) : 1; }();
    return v;
}

int gn_chunks(int n_inst, long long rows_per_inst, int c) {
    const int rpp = 256 / (c >> 3);
    const int var = gn_variant();
    const int passes = var == 0 ? 32 : var == 1 ? 16 : 8;
    long long by_rows = (rows_per_inst + (long long)rpp * passes - 1) / ((long long)rpp * passes);
    long long want = (var == 0 ? 4096 : var == 1 ? 8192 : 16384) / (n_inst > 0 ? n_inst : 1); if (want < 1) want = 1;
    long long ch = by_rows < want ? by_rows : want;
    const int cap = var == 0 ? 512 : GN_MAX_CHUNKS;
    if (ch > cap) ch = cap;
    if (ch < 1) ch = 1;
    return (int)ch;
}

}  // namespace

extern "C" int64_t uav_groupnorm_workspace_bytes(int32_t n_inst, int32_t c) {
    return (int64_t)n_inst * (gn_variant() == 0 ? 512 : GN_MAX_CHUNKS) * c * 2 * 4;
}

extern "C" int uav_groupnorm_scale_shift(const void* x1, const void* x2, int32_t x_f32, int32_t c1, int32_t c2, int64_t x2_rows,
                                         int32_t c_real,
                                         int32_t n_inst, int64_t rows_per_inst, int32_t groups, float eps,
                                         const float* gamma, const float* beta, float* scale_out, float* shift_out,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x1 || !scale_out || !shift_out || !workspace) return UAV_EINVAL;
    const int c = c1 + c2;
    if (c1 <= 0 || c2 < 0 || (c1 % 8) || (c2 % 8) || c > 2048 || (c2 > 0 && !x2)) return UAV_ESHAPE;
    if (c_real <= 0 || c_real > c || groups <= 0 || (c_real % groups) || n_inst <= 0 || rows_per_inst <= 0) return UAV_ESHAPE;
    if (n_inst > 65535) return UAV_ESHAPE;
    const int chunks = gn_chunks(n_inst, rows_per_inst, c);
    if (workspace_bytes < (int64_t)n_inst * chunks * groups * 2 * 4) return UAV_EINVAL;
    if (groups > 2048) return UAV_ESHAPE;
    if (x2_rows && (c2 <= 0 || x2_rows * 2 != (int64_t)n_inst * rows_per_inst)) return UAV_ESHAPE;
    GnSrc s{(const char*)x1, (const char*)x2, c1, c2, (long long)x2_rows};
    hipStream_t st = (hipStream_t)stream;
    const int var = gn_variant();
#define GN_PARTIAL(F, UU) hipLaunchKernelGGL((gn_partial_kernel<F, UU>), dim3(chunks, n_inst), dim3(256), 0, st, s, \
                                             (long long)rows_per_inst, chunks, c_real, groups, (float*)workspace)
    if (x_f32) { if (var == 0) GN_PARTIAL(true, 8); else GN_PARTIAL(true, 4); }
    else if (var == 0) GN_PARTIAL(false, 8);
    else if (var == 3) GN_PARTIAL(false, 8);
    else GN_PARTIAL(false, 4);
#undef GN_PARTIAL
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n_inst), dim3(256), 0, st, (const float*)workspace, chunks, c, c_real,
                       groups, (long long)rows_per_inst, eps, gamma, beta, scale_out, shift_out);
    return uav_launch_status();
}

extern "C" int uav_groupnorm_finalize_partials(const float* partials, int64_t chunks_total, int32_t chunk_rows, int32_t c,
                                               int32_t n_inst, int64_t rows_per_inst, int32_t groups, float eps,
                                               const float* gamma, const float* beta, float* scale_out, float* shift_out,
                                               void* stream) {
    if (!partials || !scale_out || !shift_out) return UAV_EINVAL;
    if (c <= 0 || groups <= 0 || groups > 2048 || (c % groups) || n_inst <= 0 || n_inst > 65535 || rows_per_inst <= 0 ||
        chunk_rows <= 0 || (rows_per_inst % chunk_rows) || (int64_t)n_inst * (rows_per_inst / chunk_rows) != chunks_total)
        return UAV_ESHAPE;
    hipLaunchKernelGGL(gn_finalize_partials_kernel, dim3(groups, n_inst), dim3(1024), 0, (hipStream_t)stream, partials,
                       (long long)chunks_total, (long long)(rows_per_inst / chunk_rows), c, groups, (long long)rows_per_inst,
                       eps, gamma, beta, scale_out, shift_out);
    return uav_launch_status();
}

extern "C" int uav_groupnorm_finalize_partials2(const float* partials1, int64_t chunks_total1, int32_t c1, int32_t groups1, int32_t n_inst1,
                                                const float* partials2, int64_t chunks_total2, int32_t c2, int32_t groups2, int32_t n_inst2,
                                                int32_t chunk_rows, int32_t n_inst, int64_t rows_per_inst, int32_t groups, float eps,
                                                const float* gamma, const float* beta, float* scale_out, float* shift_out,
                                                void* stream) {
    if (!partials1 || !partials2 || !scale_out || !shift_out) return UAV_EINVAL;
    const int c = c1 + c2;
    if (c1 <= 0 || c2 <= 0 || groups <= 0 || groups > 2048 || (c % groups) || groups1 <= 0 || groups2 <= 0 || (c1 % groups1) ||
        (c2 % groups2) || n_inst <= 0 || n_inst > 65535 || rows_per_inst <= 0 || chunk_rows <= 0 || (rows_per_inst % chunk_rows))
        return UAV_ESHAPE;
    const int cpg = c / groups;
    if ((c1 % cpg) || (cpg % (c1 / groups1)) || (cpg % (c2 / groups2))) return UAV_ESHAPE;      // groups must not straddle
    const long long chunks = rows_per_inst / chunk_rows;
    if (n_inst1 <= 0 || n_inst2 <= 0 || (n_inst % n_inst1) || (n_inst % n_inst2) || chunks_total1 != (int64_t)n_inst1 * chunks ||
        chunks_total2 != (int64_t)n_inst2 * chunks)
        return UAV_ESHAPE;
    GnSrcPartials s1{partials1, (long long)chunks_total1, chunks, c1, groups1, n_inst1};
    GnSrcPartials s2{partials2, (long long)chunks_total2, chunks, c2, groups2, n_inst2};
    hipLaunchKernelGGL(gn_finalize_partials2_kernel, dim3(groups, n_inst), dim3(1024), 0, (hipStream_t)stream, s1, s2, groups,
                       (long long)rows_per_inst, eps, gamma, beta, scale_out, shift_out);
    return uav_launch_status();
}

extern "C" int uav_groupnorm_apply(const void* x1, const void* x2, int32_t x_f32, int32_t c1, int32_t c2, int64_t x2_rows,
                                   int32_t n_inst, int64_t rows_per_inst, const float* scale, const float* shift, int32_t silu,
                                   void* y, void* raw_f16_out, int32_t raw_hilo, void* stream) {
    if (!x1 || !scale || !shift || !y) return UAV_EINVAL;
    const int c = c1 + c2;
    if (c1 <= 0 || c2 < 0 || (c1 % 8) || (c2 % 8) || c > 2048 || (c2 > 0 && !x2)) return UAV_ESHAPE;
    if (n_inst <= 0 || n_inst > 65535 || rows_per_inst <= 0) return UAV_ESHAPE;
    const int rpp = 256 / (c >> 3);
    long long chunks = (rows_per_inst + (long long)rpp * 4 - 1) / ((long long)rpp * 4);
    long long want = 8192 / n_inst; if (want < 1) want = 1;
    if (chunks > want) chunks = want;
    if (chunks < 1) chunks = 1;
    if (x2_rows && (c2 <= 0 || x2_rows * 2 != (int64_t)n_inst * rows_per_inst)) return UAV_ESHAPE;
    GnSrc s{(const char*)x1, (const char*)x2, c1, c2, (long long)x2_rows};
    if (x_f32)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)chunks, n_inst), dim3(256), 0, (hipStream_t)stream, s,
                           (long long)rows_per_inst, (int)chunks, scale, shift, silu, (char*)y, (char*)raw_f16_out, raw_hilo);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)chunks, n_inst), dim3(256), 0, (hipStream_t)stream, s,
                           (long long)rows_per_inst, (int)chunks, scale, shift, silu, (char*)y, (char*)raw_f16_out, raw_hilo);
    return uav_launch_status();
}

static int layernorm_launch(const void* x, int x_f32, void* y, const float* gamma, const float* beta, int64_t rows, int32_t c,
                            float eps, void* stream) {
    if (!x || !y || !gamma || !beta) return UAV_EINVAL;
    if (c <= 0 || (c % 8) || c > 2048 || rows <= 0) return UAV_ESHAPE;
    long long blocks = (rows + 3) / 4; if (blocks > 4096) blocks = 4096;
    const int nv = ((c >> 3) + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define LN_LAUNCH(NV, F) hipLaunchKernelGGL((layernorm_kernel<NV, F>), dim3((unsigned)blocks), dim3(256), 0, st, (const char*)x, \
                                            (char*)y, gamma, beta, (long long)rows, c, eps)
    if (x_f32) { if (nv == 1) LN_LAUNCH(1, true); else if (nv == 2) LN_LAUNCH(2, true); else if (nv == 3) LN_LAUNCH(3, true); else LN_LAUNCH(4, true); }
    else { if (nv == 1) LN_LAUNCH(1, false); else if (nv == 2) LN_LAUNCH(2, false); else if (nv == 3) LN_LAUNCH(3, false); else LN_LAUNCH(4, false); }
#undef LN_LAUNCH
    return uav_launch_status();
}

extern "C" int uav_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int32_t c,
                                 float eps, void* stream) {
    return layernorm_launch(x, 0, y, gamma, beta, rows, c, eps, stream);
}

extern "C" int uav_layernorm_f32in(const float* x, void* y, const float* gamma, const float* beta, int64_t rows, int32_t c,
                                   float eps, void* stream) {
    return layernorm_launch(x, 1, y, gamma, beta, rows, c, eps, stream);
}
