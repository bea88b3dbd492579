// K5/K6/K8 — flash attention on MFMA for gfx950: softmax(scale * Q K^T) V, fp16 in/out,
// fp32 scores / softmax / accumulation, no mask.  Replaces CrossAttention._attention
// (attention.py:209-238) for spatial self-attention (L=1600, d=128) and text cross-attention
// (Lk=77, d=64/128), and the VAE AttentionBlock core (diffusers_attention.py:341-369;
// single head d=512, L = H*W up to 102400) without materialising the L x L scores.
//
// Workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.
// Both MFMAs are issued so that a LANE owns a QUERY:
//   S^T[key][q]  = K[key][:] . Q[q][:]     (A = K fragment from LDS, B = Q fragment in VGPRs)
//   O^T[d][q]   += V^T[d][key] * P^T[key][q] (A = V^T fragment from LDS, B = P in VGPRs)
// so running max / sum / rescale are per-lane scalars and P feeds the second MFMA straight
// from the accumulator registers of the first: the k-index permutation the 32x32 C-layout
// imposes on P (keys {0-3,8-11 | 4-7,12-15} per half-wave) is mirrored when reading V^T,
// no cross-lane traffic (cdna guide T12 without the permlane).
// K tiles (32 keys) arrive by global_load_lds DMA, double buffered, XOR-swizzled on the source
// side; V tiles are register-staged and transposed 4x8 on the way into a padded V^T image
// (72-B rows: conflict-free ds_read_b64).
// Roofline: MFMA-bound for L >= ~1k (4*L*d FLOP per query row); the text cross-attention
// (Lk=77) is HBM-bound on reading Q / writing O.
#include "uav_common.h"
#include <stdlib.h>
#include <utility>

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnArgs {
    const char* q; long long q_stride;
    const char* k; long long k_stride;
    const char* v; long long v_stride;
    char* o; long long o_stride;
    int bq, lq, lk, q_per_kv, heads;
    float scale_log2;
    const char* zero_page;
    int causal;                 // 1: key j is visible to query i only if j <= i (CLIP text encoder)
};

constexpr int KV = 32;          // keys per tile
constexpr int VT_STRIDE = 72;   // bytes per V^T row (32 keys * 2 B + 8 B pad)

template <int D> struct AttnCfg {
    static constexpr int KS_BYTES = KV * D * 2;            // one K stage
    static constexpr int VT_BYTES = D * VT_STRIDE;
    static constexpr int SMEM = 2 * KS_BYTES + VT_BYTES;
    static constexpr int SLOTS = D / 8;                    // 16-B slots per K row
    static constexpr int KPIECES = KV * SLOTS / 256;       // DMA pieces per thread per tile
    static constexpr int VUNITS = (D + 255) / 256;         // (4 keys x 8 d) units per thread
};

template <int D> UAV_DEVINL int kswz(int key) { return D == 64 ? ((key >> 1) & 7) : (key & 15); }

template <int D>
__global__ __launch_bounds__(256, (D <= 128 ? 2 : 1)) void attn_kernel(AttnArgs p) {
    using C = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                       // [2][KV][D] halves, swizzled
    char* Vt = smem + 2 * C::KS_BYTES;     // [D][36] halves
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int bk = b / p.q_per_kv;
    const char* kbase = p.k + ((long long)bk * p.lk * p.k_stride + (long long)h * D) * 2;
    const char* vbase = p.v + ((long long)bk * p.lk * p.v_stride + (long long)h * D) * 2;

    // ---- Q fragments (B operand): Q[q][16s + 8hi .. +8] ------------------------------------
    const int qrow = q0 + l32;
    const int qr = qrow < p.lq ? qrow : p.lq - 1;
    const char* qptr = p.q + (((long long)b * p.lq + qr) * p.q_stride + (long long)h * D) * 2;
    half8_t qf[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) qf[s] = *(const half8_t*)(qptr + (16 * s + 8 * hi) * 2);

    float16_t oacc[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (p.lk + KV - 1) / KV;

    // ---- staging roles ----------------------------------------------------------------------
    auto issue_k = [&](int stage, int t) {
        char* dst = Ks + stage * C::KS_BYTES;
#pragma unroll
        for (int ps = 0; ps < C::KPIECES; ++ps) {
            const int pi = ps * 256 + tid;
            const int row = pi / C::SLOTS, sp = pi % C::SLOTS;
            const int sl = sp ^ kswz<D>(row);
            const int key = t * KV + row;
            const char* g = key < p.lk ? kbase + ((long long)key * p.k_stride + sl * 8) * 2 : p.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + (ps * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    half8_t vst[C::VUNITS][4];
    auto load_v = [&](int t) {
#pragma unroll
        for (int u = 0; u < C::VUNITS; ++u) {
            const int unit = u * 256 + tid;
            if (unit < D) {
                const int kg = unit & 7, dv = unit >> 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = t * KV + kg * 4 + i;
                    half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                    vst[u][i] = key < p.lk ? *(const half8_t*)(vbase + ((long long)key * p.v_stride + dv * 8) * 2) : z;
                }
            }
        }
    };
    auto store_v = [&]() {
#pragma unroll
        for (int u = 0; u < C::VUNITS; ++u) {
            const int unit = u * 256 + tid;
            if (unit < D) {
                const int kg = unit & 7, dv = unit >> 3;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    half4_t w = {vst[u][0][e], vst[u][1][e], vst[u][2][e], vst[u][3][e]};
                    *(half4_t*)(Vt + (dv * 8 + e) * VT_STRIDE + kg * 8) = w;
                }
            }
        }
    };

    issue_k(0, 0);
    load_v(0);
    for (int t = 0; t < nt; ++t) {
        store_v();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) { issue_k((t + 1) & 1, t + 1); load_v(t + 1); }

        // ---- S^T = K Q^T ---------------------------------------------------------------------
        const char* kst = Ks + (t & 1) * C::KS_BYTES + l32 * (2 * D);
        const int ksw = kswz<D>(l32);
        float16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            half8_t kf = *(const half8_t*)(kst + (((2 * s + hi) ^ ksw) << 4));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
        }
        // ---- online softmax: lane = query, registers = 16 keys --------------------------------
        const int key0 = t * KV + 4 * hi;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2);
            float s = sacc[r] * p.scale_log2;
            s = (key < p.lk) & (!p.causal | (key <= qrow)) ? s : -INFINITY;
            sacc[r] = s; mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sacc[r] - m_new);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // exact deferred rescale (cdna guide T13 with threshold 0): once the running maxima of all the
        // wave's rows have settled alpha == 1 for every lane and the O-wide multiply is skipped
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        // ---- O^T += V^T P^T -------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {
            const char* vrow = Vt + (i * 32 + l32) * VT_STRIDE + hi * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half4_t a = *(const half4_t*)(vrow + s2 * 32);
                half4_t c = *(const half4_t*)(vrow + s2 * 32 + 16);
                half8_t vf = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], oacc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.lq) {
        char* optr = p.o + (((long long)b * p.lq + qrow) * p.o_stride + (long long)h * D) * 2;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4_t o = {(half_t)(oacc[i][4 * g] * inv), (half_t)(oacc[i][4 * g + 1] * inv),
                             (half_t)(oacc[i][4 * g + 2] * inv), (half_t)(oacc[i][4 * g + 3] * inv)};
                *(half4_t*)(optr + (i * 32 + 8 * g + 4 * hi) * 2) = o;
            }
    }
}


// ---------------------------------------------------------------------------------------------
// Short key sets (text cross-attention: Lk = 77 -> 3 tiles).  attn_kernel walks its tiles with one L2 round trip per tile
// (issue tile t+1, compute tile t — which takes a fraction of a round trip —, wait, barrier): with 3 tiles a workgroup spends
// ~5.5 us for 32 KiB of Q / O traffic and the kernel sits at 3.0 TB/s.  Here EVERY K tile (LDS-DMA) and V tile (registers ->
// V^T images) is requested before anything is waited for: one round trip, one barrier, then the 3 tiles back to back.
// Same arithmetic, same order: bit-identical to attn_kernel.
constexpr int NTS = 3;
template <int D>
__global__ __launch_bounds__(256, 2) void attn_short_kernel(AttnArgs p) {
    using C = AttnCfg<D>;
    static_assert(C::VUNITS == 1, "short-key kernel: head dims up to 256");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                // [NTS][KV][D] halves, swizzled
    char* Vt = smem + NTS * C::KS_BYTES;            // [NTS][D][36] halves
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int bk = b / p.q_per_kv;
    const char* kbase = p.k + ((long long)bk * p.lk * p.k_stride + (long long)h * D) * 2;
    const char* vbase = p.v + ((long long)bk * p.lk * p.v_stride + (long long)h * D) * 2;
    const int nt = (p.lk + KV - 1) / KV;            // <= NTS (host)

    // Q rows (and, at the end, O rows) move ROW-COALESCED through a wave-private LDS image (round 5): with a lane owning a query, the 64
    // lanes of one 16-B load touch 32 rows x 32 B — quarter cache lines, the transaction-bound pattern measured on the conv epilogues
    //  — and this kernel does little else than read Q and write O (77 keys: 3.1 TB/s before).  Here D / 8 lanes read one row's
    // D halves contiguously (whole lines), the image (rows of 2 D + 16 B: conflict-free both ways) hands every lane its query's fragments.
    // d = 64 only (measured, run 29: text cross-attention 101.7 -> 94.0 ms per clip, 3.09 -> 3.35 TB/s): at d = 128 the images take the
    // second workgroup's LDS and the kernel got slower, so that instance keeps the lane-per-row accesses.
    constexpr bool XCO = D <= 64;
    constexpr int CH = D / 8;                       // 16-B chunks per row = lanes per row
    constexpr int RPI = 64 / CH;                    // rows per wave-wide access
    constexpr int XROW = 2 * D + 16;                // bytes per staged row
    char* const xw = smem + NTS * (C::KS_BYTES + C::VT_BYTES) + wave * (32 * XROW);
    const int qrow = q0 + l32;
    half8_t qf[D / 16];
    if constexpr (!XCO) {
        const int qr = qrow < p.lq ? qrow : p.lq - 1;
        const char* qptr = p.q + (((long long)b * p.lq + qr) * p.q_stride + (long long)h * D) * 2;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) qf[s] = *(const half8_t*)(qptr + (16 * s + 8 * hi) * 2);
    } else {
        const int xr = lane / CH, xc = lane % CH;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) {
            const int r = i * RPI + xr;             // row of this wave's 32
            const int qg = q0 + r < p.lq ? q0 + r : p.lq - 1;
            const half8_t v = *(const half8_t*)(p.q + (((long long)b * p.lq + qg) * p.q_stride + (long long)h * D) * 2 + xc * 16);
            *(half8_t*)(xw + r * XROW + xc * 16) = v;
        }
#pragma unroll
        for (int s = 0; s < D / 16; ++s) qf[s] = *(const half8_t*)(xw + l32 * XROW + (16 * s + 8 * hi) * 2);
    }

    // ---- all K tiles by DMA, all V tiles into registers: nothing is waited for in between -----
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
        if (t < nt) {
            char* dst = Ks + t * C::KS_BYTES;
#pragma unroll
            for (int ps = 0; ps < C::KPIECES; ++ps) {
                const int pi = ps * 256 + tid;
                const int row = pi / C::SLOTS, sp = pi % C::SLOTS;
                const int sl = sp ^ kswz<D>(row);
                const int key = t * KV + row;
                const char* g = key < p.lk ? kbase + ((long long)key * p.k_stride + sl * 8) * 2 : p.zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + (ps * 256 + wave * 64) * 16), 16, 0, 0);
            }
        }
    }
    half8_t vst[NTS][4];
    const bool vrole = tid < D;
    const int kg = tid & 7, dv = tid >> 3;
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = t * KV + kg * 4 + i;
            half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            vst[t][i] = (vrole && t < nt && key < p.lk) ? *(const half8_t*)(vbase + ((long long)key * p.v_stride + dv * 8) * 2) : z;
        }
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
        if (vrole && t < nt) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                half4_t w = {vst[t][0][e], vst[t][1][e], vst[t][2][e], vst[t][3][e]};
                *(half4_t*)(Vt + t * C::VT_BYTES + (dv * 8 + e) * VT_STRIDE + kg * 8) = w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float16_t oacc[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int t = 0; t < nt; ++t) {
        const char* kst = Ks + t * C::KS_BYTES + l32 * (2 * D);
        const int ksw = kswz<D>(l32);
        float16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            half8_t kf = *(const half8_t*)(kst + (((2 * s + hi) ^ ksw) << 4));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
        }
        const int key0 = t * KV + 4 * hi;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2);
            float s = sacc[r] * p.scale_log2;
            s = key < p.lk ? s : -INFINITY;
            sacc[r] = s; mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sacc[r] - m_new);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {
            const char* vrow = Vt + t * C::VT_BYTES + (i * 32 + l32) * VT_STRIDE + hi * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half4_t a = *(const half4_t*)(vrow + s2 * 32);
                half4_t c = *(const half4_t*)(vrow + s2 * 32 + 16);
                half8_t vf = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], oacc[i], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (!XCO) {
        if (qrow < p.lq) {
            char* optr = p.o + (((long long)b * p.lq + qrow) * p.o_stride + (long long)h * D) * 2;
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4_t o = {(half_t)(oacc[i][4 * g] * inv), (half_t)(oacc[i][4 * g + 1] * inv),
                                 (half_t)(oacc[i][4 * g + 2] * inv), (half_t)(oacc[i][4 * g + 3] * inv)};
                    *(half4_t*)(optr + (i * 32 + 8 * g + 4 * hi) * 2) = o;
                }
        }
        return;
    }
    // O rows leave the same way: the lane's 8-B pieces into the wave's image (the Q fragments are in registers since the top), whole rows out
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4_t o = {(half_t)(oacc[i][4 * g] * inv), (half_t)(oacc[i][4 * g + 1] * inv),
                         (half_t)(oacc[i][4 * g + 2] * inv), (half_t)(oacc[i][4 * g + 3] * inv)};
            *(half4_t*)(xw + l32 * XROW + (i * 32 + 8 * g + 4 * hi) * 2) = o;
        }
    {
        const int xr = lane / CH, xc = lane % CH;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) {
            const int r = i * RPI + xr;
            const half8_t v = *(const half8_t*)(xw + r * XROW + xc * 16);
            if (q0 + r < p.lq)
                *(half8_t*)(p.o + (((long long)b * p.lq + q0 + r) * p.o_stride + (long long)h * D) * 2 + xc * 16) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// head_dim = 512 (VAE mid-block attention, one head, L = H*W up to 102400; 57 % of the VAE FLOPs).
// The generic kernel above would need 256 (O^T) + 128 (Q) accumulator/operand registers per lane
// and spills.  Here a PAIR of waves shares 32 queries and splits d in halves of 256:
//   * each wave computes the partial S^T over ITS 256 dims (16 MFMAs), the two partials are
//     summed through a 4-KiB LDS exchange slot per wave, both waves then run the same (cheap,
//     per-lane) online softmax;
//   * each wave accumulates O^T for ITS 256 output dims (8 tiles x 2 k-steps = 16 MFMAs).
// 8 waves (512 threads) = 4 pairs = 128 queries per workgroup, 2 waves per SIMD at <= 256 VGPRs,
// LDS: K 2 x 32 KiB (DMA, double buffered) + V^T 36 KiB + exchange 32 KiB = 132 KiB.
constexpr int K5_BYTES = KV * 512 * 2;         // 32 KiB per K stage
constexpr int V5_BYTES = 512 * VT_STRIDE;      // 36 KiB
constexpr int X5_BYTES = 8 * 4096;             // partial-S exchange
constexpr int SMEM5 = 2 * K5_BYTES + V5_BYTES + X5_BYTES;

#ifdef UAV_DEV_KERNELS       // round-1 pair-split form (100 B of scratch): development build only (UAV_ATTN512=0), VERDICT r5 #12
__global__ __launch_bounds__(512, 2) void attn512_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + 2 * K5_BYTES;
    char* Ex = Vt + V5_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave >> 1, dh = wave & 1;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + qg * 32;
    const int bk = b / p.q_per_kv;
    const char* kbase = p.k + (long long)bk * p.lk * p.k_stride * 2;
    const char* vbase = p.v + (long long)bk * p.lk * p.v_stride * 2;

    const int qrow = q0 + l32;
    const int qr = qrow < p.lq ? qrow : p.lq - 1;
    const char* qptr = p.q + (((long long)b * p.lq + qr) * p.q_stride + dh * 256) * 2;
    half8_t qf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) qf[s] = *(const half8_t*)(qptr + (16 * s + 8 * hi) * 2);

    float16_t oacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int nt = (p.lk + KV - 1) / KV;

    auto issue_k = [&](int stage, int t) {
        char* dst = Ks + stage * K5_BYTES;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + wave;                 // 64 slots per row: one row per wave-instruction
            const int sl = lane ^ (row & 15);
            const int key = t * KV + row;
            const char* g = key < p.lk ? kbase + ((long long)key * p.k_stride + sl * 8) * 2 : p.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + (ps * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    half8_t vst[4];
    const int vkg = tid & 7, vdv = tid >> 3;
    auto load_v = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = t * KV + vkg * 4 + i;
            half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            vst[i] = key < p.lk ? *(const half8_t*)(vbase + ((long long)key * p.v_stride + vdv * 8) * 2) : z;
        }
    };
    auto store_v = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            half4_t w = {vst[0][e], vst[1][e], vst[2][e], vst[3][e]};
            *(half4_t*)(Vt + (vdv * 8 + e) * VT_STRIDE + vkg * 8) = w;
        }
    };

    issue_k(0, 0);
    load_v(0);
    char* exw = Ex + wave * 4096 + lane * 16;
    const char* exr = Ex + (wave ^ 1) * 4096 + lane * 16;
    for (int t = 0; t < nt; ++t) {
        store_v();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) { issue_k((t + 1) & 1, t + 1); load_v(t + 1); }

        const char* kst = Ks + (t & 1) * K5_BYTES + l32 * 1024;
        const int ksw = l32 & 15;
        float16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            half8_t kf = *(const half8_t*)(kst + (((dh * 32 + 2 * s + hi) ^ ksw) << 4));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
        }
        // exchange partial scores with the partner wave (other half of d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            float4_t v = {sacc[4 * r4], sacc[4 * r4 + 1], sacc[4 * r4 + 2], sacc[4 * r4 + 3]};
            *(float4_t*)(exw + r4 * 1024) = v;
        }
        __syncthreads();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            float4_t v = *(const float4_t*)(exr + r4 * 1024);
            sacc[4 * r4] += v[0]; sacc[4 * r4 + 1] += v[1]; sacc[4 * r4 + 2] += v[2]; sacc[4 * r4 + 3] += v[3];
        }
        const int key0 = t * KV + 4 * hi;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2);
            float sc = sacc[r] * p.scale_log2;
            sc = key < p.lk ? sc : -INFINITY;
            sacc[r] = sc; mx = fmaxf(mx, sc);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sacc[r] - m_new);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                           // exact deferred rescale, see attn_kernel
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* vrow = Vt + ((dh * 8 + i) * 32 + l32) * VT_STRIDE + hi * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half4_t a = *(const half4_t*)(vrow + s2 * 32);
                half4_t c = *(const half4_t*)(vrow + s2 * 32 + 16);
                half8_t vf = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], oacc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.lq) {
        char* optr = p.o + (((long long)b * p.lq + qrow) * p.o_stride + dh * 256) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4_t o = {(half_t)(oacc[i][4 * g] * inv), (half_t)(oacc[i][4 * g + 1] * inv),
                             (half_t)(oacc[i][4 * g + 2] * inv), (half_t)(oacc[i][4 * g + 3] * inv)};
                *(half4_t*)(optr + (i * 32 + 8 * g + 4 * hi) * 2) = o;
            }
    }
}

#endif  // UAV_DEV_KERNELS
// ---------------------------------------------------------------------------------------------
// head_dim = 512, ONE WAVE PER SIMD (round 2).  The pair-split kernel above spends three barriers and a 32-KiB
// partial-score exchange per 32-key tile because two waves share each query block.  Here a wave owns 32 queries and the
// WHOLE head: 128 Q registers in the VGPR half and the 256 fp32 O^T accumulators in the ACCUMULATOR half of the unified
// register file (a 256-thread workgroup at one wave per SIMD may use all 512) -> no exchange, no pair, and with the V^T
// image double-buffered ONE barrier per tile.  4 waves = 128 queries per workgroup; per tile a wave issues 32 MFMAs for
// S^T = K Q^T and 32 for O^T += V^T P^T, fed from the same K (DMA) / V^T (register-transposed) LDS images as before.
// hipcc cannot allocate this itself (with the accumulators as C++ variables it shuffled them between the two halves:
// 1306 v_accvgpr moves and 377 spilled registers in the loop), so O^T tile i lives in a[16i : 16i+15] BY NAME: the P.V
// MFMAs, the (rare) online-softmax rescale and the final read-out are inline asm on those registers; every statement
// lists the whole accumulator file as clobbered, which also makes the kernel descriptor allocate it, and the compiler
// never touches AGPRs itself (checked in the .s: no v_accvgpr outside the asm blocks, no scratch).
#define O5W_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
    "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int I> UAV_DEVINL void o5w_mfma(const half8_t& va, const half8_t& pb) {
    // s_nop 1: the V^T fragment may have been packed by VALU moves right before (VALU write -> MFMA operand read)
    asm volatile("s_nop 1\n v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(va), "v"(pb), "i"(16 * I), "i"(16 * I + 15) : O5W_CLOBBERS);
}
// S^T accumulation in VGPRs BY CONSTRAINT: left to itself hipcc put this chain into a[0:15] — on top of O^T tile 0.
// S^T chain.  `pending` = LDS reads issued after this step's K fragment (s_waitcnt lgkmcnt takes an immediate).
UAV_DEVINL void o5w_kread(half8_t& kf, unsigned addr, int off) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf) : "v"(addr), "n"(off));
}
UAV_DEVINL void o5w_mfma_s0(float16_t& acc, const half8_t& a, const half8_t& b, int pending) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b), "n"(pending));
}
UAV_DEVINL void o5w_mfma_s(float16_t& acc, const half8_t& a, const half8_t& b, int pending) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b), "n"(pending));
}
template <int N> UAV_DEVINL void o5w_zero1() { asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "i"(N) : O5W_CLOBBERS); }
template <int N> UAV_DEVINL void o5w_scale1(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n s_nop 1\n v_mul_f32 %0, %0, %1\n s_nop 1\n v_accvgpr_write_b32 a%c2, %0"
                 : "=&v"(t) : "v"(f), "i"(N) : O5W_CLOBBERS);
}
template <int N> UAV_DEVINL float o5w_read1() { float t; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(t) : "i"(N)); return t; }
template <int... N> UAV_DEVINL void o5w_zero_all(std::integer_sequence<int, N...>) { (o5w_zero1<N>(), ...); }
template <int... N> UAV_DEVINL void o5w_scale_all(float f, std::integer_sequence<int, N...>) { (o5w_scale1<N>(f), ...); }
template <int B, int... N> UAV_DEVINL void o5w_read16(float (&o)[16], std::integer_sequence<int, N...>) { ((o[N] = o5w_read1<B + N>()), ...); }

constexpr int SMEM5W = 2 * K5_BYTES + 2 * V5_BYTES;      // 64 KiB K + 72 KiB V^T = 136 KiB

// O^T += V^T P^T over the 16 d-tiles.  The P.V MFMAs are volatile asm (named accumulators), i.e. barriers for the
// scheduler: a V^T fragment read placed right before its MFMA is waited for in full.  The reads are therefore written
// VPF tiles AHEAD of their use in source order (ring of VPF fragment pairs, 8 VGPRs each); the compiler's counted
// lgkmcnt keeps VPF-1 tiles of reads in flight behind every MFMA pair.
constexpr int VPF = 3;
UAV_DEVINL void o5w_v_frag(half8_t (&vf)[2], const char* vt, int tile, int l32, int hi) {
    const char* vrow = vt + (tile * 32 + l32) * VT_STRIDE + hi * 8;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        half4_t a = *(const half4_t*)(vrow + s2 * 32);
        half4_t c = *(const half4_t*)(vrow + s2 * 32 + 16);
        vf[s2] = half8_t{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    }
}
template <int I> UAV_DEVINL void o5w_pv_tile(half8_t (&ring)[VPF][2], const char* vt, int l32, int hi, const half8_t (&pf)[2]) {
    o5w_mfma<I>(ring[I % VPF][0], pf[0]);
    o5w_mfma<I>(ring[I % VPF][1], pf[1]);
    if (I + VPF < 16) o5w_v_frag(ring[I % VPF], vt, I + VPF, l32, hi);
}
template <int... I> UAV_DEVINL void o5w_pv_all(const char* vt, int l32, int hi, const half8_t (&pf)[2], std::integer_sequence<int, I...>) {
    half8_t ring[VPF][2];
#pragma unroll
    for (int i = 0; i < VPF; ++i) o5w_v_frag(ring[i], vt, i, l32, hi);
    (o5w_pv_tile<I>(ring, vt, l32, hi, pf), ...);
}
template <int I> UAV_DEVINL void o5w_store_tile(char* optr, int hi, float inv) {
    float o[16];
    o5w_read16<16 * I>(o, std::make_integer_sequence<int, 16>{});
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        half4_t h = {(half_t)(o[4 * g] * inv), (half_t)(o[4 * g + 1] * inv), (half_t)(o[4 * g + 2] * inv), (half_t)(o[4 * g + 3] * inv)};
        *(half4_t*)(optr + (I * 32 + 8 * g + 4 * hi) * 2) = h;
    }
}
template <int... I> UAV_DEVINL void o5w_store_all(char* optr, int hi, float inv, std::integer_sequence<int, I...>) {
    (o5w_store_tile<I>(optr, hi, inv), ...);
}

__global__ __launch_bounds__(256, 1) void attn512w_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + 2 * K5_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int bk = b / p.q_per_kv;
    const char* kbase = p.k + (long long)bk * p.lk * p.k_stride * 2;
    const char* vbase = p.v + (long long)bk * p.lk * p.v_stride * 2;

    const int qrow = q0 + l32;
    const int qr = qrow < p.lq ? qrow : p.lq - 1;
    const char* qptr = p.q + ((long long)b * p.lq + qr) * p.q_stride * 2;
    half8_t qf[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) qf[s] = *(const half8_t*)(qptr + (16 * s + 8 * hi) * 2);

    const unsigned ks_lds = (unsigned)(size_t)(lptr_t)Ks + l32 * 1024;         // LDS address of this lane's key row, stage 0
    unsigned kx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kx[j] = (unsigned)(((2 * j + hi) ^ (l32 & 15)) << 4);

    o5w_zero_all(std::make_integer_sequence<int, 256>{});
    float m_run = -INFINITY, l_run = 0.f;
    const int nt = (p.lk + KV - 1) / KV;

    // Global addresses as (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane offset): the per-lane 64-bit pointers
    // of all 16 loads of a tile were ~32 loop-invariant VGPRs the Q fragments (128) and the prefetch rings have no room for.
    unsigned kvo[4];                                        // swizzled 16-B slot of this lane inside key row (4j + wave)
#pragma unroll
    for (int j = 0; j < 4; ++j) kvo[j] = (unsigned)((lane ^ ((j * 4 + wave) & 15)) * 16);
    auto issue_k = [&](int stage, int t) {                  // one 1-KiB key row per wave-instruction, 8 rows per wave
        char* dst = Ks + stage * K5_BYTES;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 4 + wave;
            const int key = t * KV + row;                   // wave-uniform
            const bool ok = key < p.lk;
            const char* base = ok ? kbase + (long long)key * p.k_stride * 2 : p.zero_page;
            const unsigned off = ok ? kvo[ps & 3] : 0u;
            __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(dst + (ps * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    half8_t vst[2][4];                                      // two (4 keys x 8 dims) units per thread
    const int vkg = tid & 7;                                // unit = u*256 + tid: key group unit & 7, dim vector unit >> 3
    const unsigned vvo = (unsigned)((vkg * 4 * (int)p.v_stride + (tid >> 3) * 8) * 2);
    auto load_v = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* base = vbase + (long long)(t * KV + i) * p.v_stride * 2;       // wave-uniform
            const bool ok = t * KV + vkg * 4 + i < p.lk;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                vst[u][i] = ok ? *(const half8_t*)(base + vvo + u * 512) : z;
            }
        }
    };
    auto store_v = [&](int buf) {
        char* vt = Vt + buf * V5_BYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int unit = u * 256 + tid, kg = unit & 7, dv = unit >> 3;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                half4_t w = {vst[u][0][e], vst[u][1][e], vst[u][2][e], vst[u][3][e]};
                *(half4_t*)(vt + (dv * 8 + e) * VT_STRIDE + kg * 8) = w;
            }
        }
    };

    issue_k(0, 0);
    load_v(0);
    for (int t = 0; t < nt; ++t) {
        store_v(t & 1);                                     // V^T buffer t&1 was last read two tiles ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // the only barrier of the tile
        if (t + 1 < nt) issue_k((t + 1) & 1, t + 1);

        // S^T = K Q^T: 32 chained MFMAs, one K fragment each.  The fragment reads run KPF steps ahead of their MFMA and
        // both are asm so that the wait is the exact count (the compiler's own bookkeeping put a full lgkmcnt(0) in front
        // of every fourth MFMA).  Fragment (key l32, dims 16s + 8hi ..) sits in 16-B slot (2s + hi) ^ (l32 & 15) of the key's
        // 1-KiB row: the XOR only touches the low 4 bits, so slot = 16 * (s >> 3) + ((2 (s & 7) + hi) ^ ksw) — 8 per-lane
        // addresses and an immediate offset instead of 32 per-lane addresses.
        const unsigned kb = ks_lds + (t & 1) * K5_BYTES;
        unsigned ka[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ka[j] = kb + kx[j];
        float16_t sacc;
        constexpr int KPF = 8;                                  // K fragments in flight ahead of the MFMA chain
        half8_t kf[KPF];
#pragma unroll
        for (int s = 0; s < KPF; ++s) o5w_kread(kf[s], ka[s & 7], (s >> 3) * 256);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int pending = (32 - s < KPF ? 32 - s : KPF) - 1;      // reads issued after this step's fragment
            if (s == 0) o5w_mfma_s0(sacc, kf[0], qf[0], pending); else o5w_mfma_s(sacc, kf[s % KPF], qf[s], pending);
            if (s + KPF < 32) o5w_kread(kf[s % KPF], ka[(s + KPF) & 7], ((s + KPF) >> 3) * 256);
        }
        // next tile's V rows (register-staged for the transposing LDS write at the top of the next iteration): issued
        // here, not beside the K DMA, so that their 32 VGPRs are not live through the S phase next to Q, the K ring and S^T
        if (t + 1 < nt) load_v(t + 1);
        asm volatile("s_nop 15" : "+v"(sacc));                // XDL write -> VALU read of the scores (hipcc cannot see the MFMAs)
        const int key0 = t * KV + 4 * hi;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2);
            float sc = sacc[r] * p.scale_log2;
            sc = key < p.lk ? sc : -INFINITY;
            sacc[r] = sc; mx = fmaxf(mx, sc);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sacc[r] - m_new);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                           // exact deferred rescale (see attn_kernel); rare after the first tiles
            asm volatile("s_nop 15\ns_nop 15" ::: "memory");   // the previous tile's MFMAs must have written the accumulators
            o5w_scale_all(alpha, std::make_integer_sequence<int, 256>{});
            asm volatile("s_nop 3" ::: "memory");
        }
        o5w_pv_all(Vt + (t & 1) * V5_BYTES, l32, hi, pf, std::make_integer_sequence<int, 16>{});
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");           // last MFMAs -> accumulator reads
    if (qrow < p.lq) {
        char* optr = p.o + ((long long)b * p.lq + qrow) * p.o_stride * 2;
        o5w_store_all(optr, hi, inv, std::make_integer_sequence<int, 16>{});
    }
}

int launch_attn512(const AttnArgs& a, hipStream_t s) {
    // one wave per SIMD (attn512w) is the production kernel at every size since its fragment reads run ahead of the MFMAs:
    // 658 vs 407 TFLOP/s at L = 102 400, 428 vs 339 at L = 25 600 (200 workgroups on 256 CUs),
    // profiles/r02_ab_attn512_fragment_prefetch_run26.log; UAV_ATTN512=0 keeps the round-1 pair-split kernel reachable for A/B
#ifdef UAV_DEV_KERNELS
    static const int variant = [] { const char* e = getenv("UAV_ATTN512"); return e ? atoi(e) : 1; }();
    if (variant == 0) {
        static UavDynLds lds;
        if (int rc = uav_set_dyn_lds(lds, (const void*)attn512_kernel, SMEM5)) return rc;
        dim3 grid((a.lq + 127) / 128, 1, a.bq);
        hipLaunchKernelGGL(attn512_kernel, grid, dim3(512), SMEM5, s, a);
        return uav_launch_status();
    }
#endif
    static UavDynLds ldsw;
    if (int rc = uav_set_dyn_lds(ldsw, (const void*)attn512w_kernel, SMEM5W)) return rc;
    hipLaunchKernelGGL(attn512w_kernel, dim3((a.lq + 127) / 128, 1, a.bq), dim3(256), SMEM5W, s, a);
    return uav_launch_status();
}

template <int D>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    using C = AttnCfg<D>;
    if constexpr (D <= 256) {
        static const int short_on = [] { const char* e = getenv("UAV_ATTN_SHORT"); return e ? atoi(e) : 1; }();   // 0: A/B
        // (d = 64: its row-coalesced O stores are 16-B pieces: rows of o_stride % 8 == 0 halves on a 16-B aligned base, else the generic kernel)
        if (short_on && a.lk <= NTS * KV && !a.causal && (D > 64 || (!(a.o_stride % 8) && !((size_t)a.o & 15)))) {
            constexpr int smem = NTS * (C::KS_BYTES + C::VT_BYTES) + (D <= 64 ? 4 * 32 * (2 * D + 16) : 0);      // + the four waves' Q / O row images (d = 64)
            static UavDynLds lds_s;
            if (smem > 65536)
                if (int rc = uav_set_dyn_lds(lds_s, (const void*)attn_short_kernel<D>, smem)) return rc;
            hipLaunchKernelGGL(attn_short_kernel<D>, dim3((a.lq + 127) / 128, a.heads, a.bq), dim3(256), smem, s, a);
            return uav_launch_status();
        }
    }
    static UavDynLds lds;
    if (C::SMEM > 65536)
        if (int rc = uav_set_dyn_lds(lds, (const void*)attn_kernel<D>, C::SMEM)) return rc;
    dim3 grid((a.lq + 127) / 128, a.heads, a.bq);
    hipLaunchKernelGGL(attn_kernel<D>, grid, dim3(256), C::SMEM, s, a);
    return uav_launch_status();
}

}  // namespace

extern "C" int uav_attention_f16(const void* q, int64_t q_stride, const void* k, int64_t k_stride, const void* v,
                                 int64_t v_stride, void* out, int64_t o_stride, int32_t bq, int32_t lq, int32_t lk,
                                 int32_t q_per_kv, int32_t heads, int32_t head_dim, float scale, int32_t causal,
                                 const void* zero_page, void* stream) {
    if (!q || !k || !v || !out || !zero_page) return UAV_EINVAL;
    if (bq <= 0 || lq <= 0 || lk <= 0 || heads <= 0 || q_per_kv <= 0 || (bq % q_per_kv)) return UAV_ESHAPE;
    if ((q_stride % 8) || (k_stride % 8) || (v_stride % 8) || (o_stride % 4)) return UAV_EALIGN;
    if (heads > 65535 || bq > 65535) return UAV_ESHAPE;
    AttnArgs a{(const char*)q, q_stride, (const char*)k, k_stride, (const char*)v, v_stride, (char*)out, o_stride,
               bq, lq, lk, q_per_kv, heads, scale * 1.44269504088896341f, (const char*)zero_page, causal ? 1 : 0};
    if (causal && (head_dim == 512 || lq != lk)) return UAV_ESHAPE;      // causal mask: self-attention in the generic kernel only
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 64: return launch_attn<64>(a, s);
        case 128: return launch_attn<128>(a, s);
#ifdef UAV_DEV_KERNELS
        case 512: return heads == 1 ? launch_attn512(a, s) : launch_attn<512>(a, s);      // attn_kernel<512>: 1 160 B of scratch, development only
#else
        case 512: return heads == 1 ? launch_attn512(a, s) : UAV_ESHAPE;                  // d = 512 is the VAE's single-head attention
#endif
        default: return UAV_ESHAPE;
    }
}
