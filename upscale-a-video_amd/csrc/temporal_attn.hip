// K7 — per-pixel temporal attention for gfx950.  Replaces TemporalAttention._attention
// (attention.py:699-733) + RelativePositionBias (attention.py:735-772) + RotaryEmbedding(32)
// (rotary-embedding-torch 0.2.3; call site attention.py:709-711):
//   q*scale -> RoPE(first rot_dim dims of each head, interleaved pairs) ; k -> RoPE
//   s_ij = q_i . k_j + bias[h][i][j] ; s -= max_j ; p = softmax_j ; out_i = sum_j p_ij v_j
// over the T (<= 8) tokens of one (batch, pixel), all heads.
//
// The reference materialises (b f) d c -> (b d) f c transposes around this op
// (attention.py:555,560).  Here the channels-last tensor is read in place: token (b,t,p) is row
// (b*T + t)*hw + p of the fused qkv projection [rows][3C].  One wave handles one pixel and a
// 512-channel group: lane = 8 consecutive channels (16-B loads, 1 KiB per wave-instruction,
// fully coalesced); the d/8 lanes of a head all-reduce the partial dot products.
// HBM-bound by construction (~4 FLOP/B): algorithmic bytes = 8*C per token (read q,k,v,
// write out, fp16).
#include "uav_common.h"
#include <stdlib.h>

namespace {

constexpr int TMAX = 8;

struct TAttnArgs {
    const char* qkv; char* out;
    int n_batch, t_len; long long hw; int c, heads, d;
    float scale; const float* rope_cos; const float* rope_sin; int rot_dim; const float* bias;
    int groups;   // ceil(c / 512)
};

// Sum over the LPH = d/8 lanes of one head (aligned lane groups) on the DPP network: quad_perm xor-1 / xor-2, then
// row_half_mirror (lane i <-> 7-i) and row_mirror (i <-> 15-i) bring in the other quad / half row — single v_add_f32_dpp
// each.  (The first version looped `__shfl_xor` over a run-time lane count: 192 dependent ds_bpermute_b32 per pixel, which
// made this HBM-class kernel latency-bound at 2 TB/s.)
template <int CTRL> UAV_DEVINL float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true);
    return x + __builtin_bit_cast(float, y);
}
template <int LPH> UAV_DEVINL float head_allreduce(float x) {
    if (LPH >= 2) x = dpp_add<0xB1>(x);          // quad_perm [1,0,3,2]
    if (LPH >= 4) x = dpp_add<0x4E>(x);          // quad_perm [2,3,0,1]
    if (LPH >= 8) x = dpp_add<0x141>(x);         // row_half_mirror
    if (LPH >= 16) x = dpp_add<0x140>(x);        // row_mirror
    if (LPH >= 32) x += __shfl_xor(x, 16, 64);
    if (LPH >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}

template <int LPH>
__global__ __launch_bounds__(256) void temporal_attn_kernel(TAttnArgs p) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwork = (long long)p.n_batch * p.hw * p.groups;
    if (wid >= nwork) return;
    const int grp = (int)(wid % p.groups);
    const long long bp = wid / p.groups;
    const long long pix = bp % p.hw;
    const int b = (int)(bp / p.hw);
    const int c0 = grp * 512 + lane * 8;
    // lanes past C keep running on clamped addresses (DPP reductions need the whole head group active); they only skip the store
    const bool live = c0 < p.c;
    const int cc = live ? c0 : 0;
    const int head = cc / p.d;
    const int sub = (cc % p.d) >> 3;          // this lane's 8-dim slice inside the head
    const bool rot = sub * 8 < p.rot_dim;
    const int T = p.t_len;
    const long long row_stride = 3ll * p.c * 2;       // bytes per qkv row
    const char* base = p.qkv + ((long long)b * T * p.hw + pix) * row_stride + (long long)cc * 2;
    const long long fstride = p.hw * row_stride;      // bytes between frames of this pixel
    const int rsub = rot ? sub * 4 : 0;

    // ---- K rotated (fp32 math, stored fp16 like the reference's fp16 rotary output), V in fp32 -------------------
    half8_t kh[TMAX];
    float vf[TMAX][8];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        const int jc = j < T ? j : T - 1;                 // rows past T are never used: read a valid one
        half8_t kx = *(const half8_t*)(base + jc * fstride + (long long)p.c * 2);
        half8_t vx = *(const half8_t*)(base + jc * fstride + (long long)p.c * 4);
        if (rot) {
            const float4_t cs = *(const float4_t*)(p.rope_cos + jc * (p.rot_dim >> 1) + rsub);
            const float4_t sn = *(const float4_t*)(p.rope_sin + jc * (p.rot_dim >> 1) + rsub);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = (float)kx[2 * q], bb = (float)kx[2 * q + 1];
                kx[2 * q] = (half_t)(a * cs[q] - bb * sn[q]);
                kx[2 * q + 1] = (half_t)(bb * cs[q] + a * sn[q]);
            }
        }
        kh[j] = kx;
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[j][e] = (float)vx[e];
    }

    char* obase = p.out + (((long long)b * T * p.hw + pix) * p.c + cc) * 2;
    const long long ofstride = p.hw * (long long)p.c * 2;
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        if (i >= T) break;
        half8_t qx = *(const half8_t*)(base + i * fstride);
        float qf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)qx[e] * p.scale;
        if (rot) {
            const float4_t cs = *(const float4_t*)(p.rope_cos + i * (p.rot_dim >> 1) + rsub);
            const float4_t sn = *(const float4_t*)(p.rope_sin + i * (p.rot_dim >> 1) + rsub);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = qf[2 * q], bb = qf[2 * q + 1];
                qf[2 * q] = a * cs[q] - bb * sn[q];
                qf[2 * q + 1] = bb * cs[q] + a * sn[q];
            }
        }
        half2_t qh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) qh[q] = half2_t{(half_t)qf[2 * q], (half_t)qf[2 * q + 1]};
        // bias row of this head: T <= 8 floats
        const float* brow = p.bias + ((long long)head * T + i) * T;
        float s[TMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) a = __builtin_amdgcn_fdot2(qh[q], half2_t{kh[j][2 * q], kh[j][2 * q + 1]}, a, false);
            a = head_allreduce<LPH>(a);
            if (j < T) { a += brow[j]; mx = fmaxf(mx, a); }
            s[j] = a;
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            s[j] = j < T ? __expf(s[j] - mx) : 0.f;
            den += s[j];
        }
        const float inv = 1.0f / den;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            const float pj = s[j] * inv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += pj * vf[j][e];
        }
        half8_t oh;
#pragma unroll
        for (int e = 0; e < 8; ++e) oh[e] = (half_t)o[e];
        if (live) *(half8_t*)(obase + i * ofstride) = oh;
    }
}


// Round 3 form of the same computation (bit-identical results).  The kernel above reads K / V frame by frame behind a
// divergent `if (rot)` block and fetches the RoPE tables and the 8 bias scalars of a row with one waited-for load each:
// ~90 SERIALIZED memory round trips per wave (ISA: `L [vmcnt 0]` 70 times) — latency-bound at 3.0 TB/s although its
// VALU work would allow ~11 TB/s.  Here
//   * the small tables (bias [heads][T][T], cos / sin [T][rot/2]) are staged in LDS once per workgroup;
//   * all 24 row loads of a pixel (q, k, v of 8 frames: 24 KiB per wave) are issued back to back, branch-free;
//   * lanes outside the rotary dims rotate by (cos, sin) = (1, 0), which is exact, instead of branching.
constexpr int TA_BIAS_MAX = 512, TA_ROPE_MAX = 256;
template <int LPH>
__global__ __launch_bounds__(256) void temporal_attn2_kernel(TAttnArgs p) {
    __shared__ float s_bias[TA_BIAS_MAX];
    __shared__ __attribute__((aligned(16))) float s_cs[TA_ROPE_MAX];
    __shared__ __attribute__((aligned(16))) float s_sn[TA_ROPE_MAX];
    const int T = p.t_len;
    const int hr = p.rot_dim >> 1;
    const int lane = threadIdx.x & 63;
    const long long nwork = (long long)p.n_batch * p.hw * p.groups;
    long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool active = wid < nwork;                 // tail waves load a valid pixel, take part in the barrier, store nothing
    if (!active) wid = nwork - 1;
    const int grp = (int)(wid % p.groups);
    const long long bp = wid / p.groups;
    const long long pix = bp % p.hw;
    const int b = (int)(bp / p.hw);
    const int c0 = grp * 512 + lane * 8;
    const bool live = c0 < p.c;
    const int cc = live ? c0 : 0;
    const int head = cc / p.d;
    const int sub = (cc % p.d) >> 3;
    const bool rot = sub * 8 < p.rot_dim;
    const long long row_stride = 3ll * p.c * 2;
    const char* base = p.qkv + ((long long)b * T * p.hw + pix) * row_stride + (long long)cc * 2;
    const long long fstride = p.hw * row_stride;
    const int rsub = rot ? sub * 4 : 0;

    // the 24 streaming loads go out first; the table staging (L2 hits) and the barrier ride in their shadow
    half8_t qx[TMAX], kh[TMAX], vx[TMAX];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        const int jc = j < T ? j : T - 1;
        qx[j] = *(const half8_t*)(base + jc * fstride);
        kh[j] = *(const half8_t*)(base + jc * fstride + (long long)p.c * 2);
        vx[j] = *(const half8_t*)(base + jc * fstride + (long long)p.c * 4);
    }
    {   // straight-line staging: 4 independent (clamped) loads per thread, one wait, then the LDS writes
        const int nb = p.heads * T * T, nr = T * hr, t0 = threadIdx.x;
        const float b0 = p.bias[t0 < nb ? t0 : 0], b1 = p.bias[t0 + 256 < nb ? t0 + 256 : 0];
        const float c_ = nr ? p.rope_cos[t0 < nr ? t0 : 0] : 0.f, s_ = nr ? p.rope_sin[t0 < nr ? t0 : 0] : 0.f;
        if (t0 < nb) s_bias[t0] = b0;
        if (t0 + 256 < nb) s_bias[t0 + 256] = b1;
        if (t0 < nr) { s_cs[t0] = c_; s_sn[t0] = s_; }
    }
    __syncthreads();
    if (!active) return;
    float vf[TMAX][8];
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        const int jc = j < T ? j : T - 1;
        float4_t cs = {1.f, 1.f, 1.f, 1.f}, sn = {0.f, 0.f, 0.f, 0.f};
        if (rot) { cs = *(const float4_t*)(s_cs + jc * hr + rsub); sn = *(const float4_t*)(s_sn + jc * hr + rsub); }
        half8_t kx = kh[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = (float)kx[2 * q], bb = (float)kx[2 * q + 1];
            kx[2 * q] = (half_t)(a * cs[q] - bb * sn[q]);
            kx[2 * q + 1] = (half_t)(bb * cs[q] + a * sn[q]);
        }
        kh[j] = kx;
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[j][e] = (float)vx[j][e];
    }
    char* obase = p.out + (((long long)b * T * p.hw + pix) * p.c + cc) * 2;
    const long long ofstride = p.hw * (long long)p.c * 2;
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        if (i >= T) break;
        float qf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)qx[i][e] * p.scale;
        float4_t cs = {1.f, 1.f, 1.f, 1.f}, sn = {0.f, 0.f, 0.f, 0.f};
        if (rot) { cs = *(const float4_t*)(s_cs + i * hr + rsub); sn = *(const float4_t*)(s_sn + i * hr + rsub); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = qf[2 * q], bb = qf[2 * q + 1];
            qf[2 * q] = a * cs[q] - bb * sn[q];
            qf[2 * q + 1] = bb * cs[q] + a * sn[q];
        }
        half2_t qh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) qh[q] = half2_t{(half_t)qf[2 * q], (half_t)qf[2 * q + 1]};
        const float* brow = s_bias + (head * T + i) * T;
        float sc[TMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) a = __builtin_amdgcn_fdot2(qh[q], half2_t{kh[j][2 * q], kh[j][2 * q + 1]}, a, false);
            a = head_allreduce<LPH>(a);
            if (j < T) { a += brow[j]; mx = fmaxf(mx, a); }
            sc[j] = a;
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            sc[j] = j < T ? __expf(sc[j] - mx) : 0.f;
            den += sc[j];
        }
        const float inv = 1.0f / den;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            const float pj = sc[j] * inv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += pj * vf[j][e];
        }
        half8_t oh;
#pragma unroll
        for (int e = 0; e < 8; ++e) oh[e] = (half_t)o[e];
        if (live) *(half8_t*)(obase + i * ofstride) = oh;
    }
}

}  // namespace

extern "C" int uav_temporal_attention_f16(const void* qkv, void* out, int32_t n_batch, int32_t t_len, int64_t hw,
                                          int32_t c, int32_t heads, float scale, const float* rope_cos,
                                          const float* rope_sin, int32_t rot_dim, const float* bias, void* stream) {
    if (!qkv || !out || !bias) return UAV_EINVAL;
    if (n_batch <= 0 || t_len <= 0 || t_len > TMAX || hw <= 0 || c <= 0 || heads <= 0 || (c % heads)) return UAV_ESHAPE;
    const int d = c / heads;
    if ((d % 8) || (d & (d - 1)) || d > 512 || (512 % d)) return UAV_ESHAPE;   // head inside one 512-channel group
    if (rot_dim < 0 || (rot_dim % 8) || rot_dim > d) return UAV_ESHAPE;
    if (rot_dim > 0 && (!rope_cos || !rope_sin)) return UAV_EINVAL;
    TAttnArgs a{(const char*)qkv, (char*)out, n_batch, t_len, (long long)hw, c, heads, d, scale,
                rope_cos, rope_sin, rot_dim, bias, (c + 511) / 512};
    const long long nwork = (long long)n_batch * hw * a.groups;
    const long long blocks = (nwork + 3) / 4;
    if (blocks >= (1ll << 31)) return UAV_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    static const int variant = [] { const char* e = getenv("UAV_TATTN"); return e ? atoi(e) : 2; }();   // 1: round-1/2 kernel (A/B)
    if (variant == 2 && heads * t_len * t_len <= TA_BIAS_MAX && t_len * (rot_dim >> 1) <= TA_ROPE_MAX && !(rot_dim & 7)) {
        switch (d >> 3) {
            case 1: hipLaunchKernelGGL(temporal_attn2_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL(temporal_attn2_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL(temporal_attn2_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
            case 8: hipLaunchKernelGGL(temporal_attn2_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
            case 16: hipLaunchKernelGGL(temporal_attn2_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
            case 32: hipLaunchKernelGGL(temporal_attn2_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL(temporal_attn2_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        }
        return uav_launch_status();
    }
    switch (d >> 3) {                       // lanes per head
        case 1: hipLaunchKernelGGL(temporal_attn_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(temporal_attn_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(temporal_attn_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL(temporal_attn_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL(temporal_attn_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL(temporal_attn_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(temporal_attn_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
    }
    return uav_launch_status();
}
