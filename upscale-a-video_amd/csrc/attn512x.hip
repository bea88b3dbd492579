// Single-head d = 512 attention of the VAE decoder's mid block (reference vae.py AttentionBlock: softmax(Q K^T / sqrt(512)) V over the
// L = H W pixels of a frame; 4 L^2 512 FLOP per frame, 21.5 TFLOP at 320 x 320) on the machinery of the fused transformer kernels
// (xattn_common.h): lane = query for the whole kernel, K and V^T PRE-PACKED as 1-KiB MFMA A fragments (attn512x_pack_kernel: once per
// frame, 0.2 GB) and streamed through the 4 x 32 KiB LDS ring by LDS-DMA, the pieces of the group three ahead between the MFMAs of the
// current one, two barriers per 32-key tile and no drain of the memory pipe; O^T (512 channels x 32 queries per wave) in the 256 named
// accumulators.  attn512w_kernel (attention.hip, round 2-5) stages K by DMA but transposes V through registers into LDS every tile, waits
// for vmcnt(0) at its one barrier and keeps eight K fragments in flight by hand: 6.3 k cycles per tile against 2.0 k of MFMA.
//
//   per wave and 32-key tile: S^T[32 keys][32 q] = K_t Q^T (32 MFMAs over d = 512, even / odd k-steps on two accumulators) -> online softmax
//   over the lane's 16 keys (+ lane ^ 32) -> P fp16 = the two B fragments of O^T += V_t^T P^T (32 MFMAs onto the 16 channel tiles, XG_WD32).
#include "xattn_common.h"

namespace {

#define XG_SA \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XS0(t0, c0, b0, 5, 6144) \
    XS0(t1, c1, b1, 5, 7168) XS(t2, c0, b2, 5, 8192) XS(t3, c1, b3, 5, 9216) XD(0, 0) XS(t4, c0, b4, 5, 10240) \
    XS(t5, c1, b5, 5, 11264) XS(t0, c0, b6, 5, 12288) XS(t1, c1, b7, 5, 13312) XD(0, 1024) XS(t2, c0, b8, 5, 14336) \
    XS(t3, c1, b9, 5, 15360) XT(t4, c0, b10, 5) XT(t5, c1, b11, 4) XD(0, 2048) XT(t0, c0, b12, 3) XT(t1, c1, b13, 2) \
    XT(t2, c0, b14, 1) XT(t3, c1, b15, 0) XD(0, 3072) XDADV

#define XG_SB \
    XRD(t0, 16384) XRD(t1, 17408) XRD(t2, 18432) XRD(t3, 19456) XRD(t4, 20480) XRD(t5, 21504) XS(t0, c0, b0, 5, 22528) \
    XS(t1, c1, b1, 5, 23552) XS(t2, c0, b2, 5, 24576) XS(t3, c1, b3, 5, 25600) XD(4096, 0) XS(t4, c0, b4, 5, 26624) \
    XS(t5, c1, b5, 5, 27648) XS(t0, c0, b6, 5, 28672) XS(t1, c1, b7, 5, 29696) XD(4096, 1024) XS(t2, c0, b8, 5, 30720) \
    XS(t3, c1, b9, 5, 31744) XT(t4, c0, b10, 5) XT(t5, c1, b11, 4) XD(4096, 2048) XT(t0, c0, b12, 3) XT(t1, c1, b13, 2) \
    XT(t2, c0, b14, 1) XT(t3, c1, b15, 0) XD(4096, 3072) XNOP

struct Attn512xArgs {
    const half_t* q; long long q_stride; const char* kp; const char* vp; half_t* o; long long o_stride;
    int lq, lk, nt; float scale_log2;
};

__global__ __launch_bounds__(256, 1) void attn512x_kernel(Attn512xArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const unsigned voff = (unsigned)(wave * XPPW * XFRAG + lane * 16);
    const long long bbytes = (long long)p.nt * XGROUP;       // packed bytes per batch entry (K and V^T each)
    const uint4_t ksrd = make_srd(p.kp + b * bbytes, (unsigned)bbytes), vsrd = make_srd(p.vp + b * bbytes, (unsigned)bbytes);
    const int ngroups = 2 * p.nt;
    auto next_of = [&](int s) -> XNext {                    // group s: tile s >> 1, K (even) or V^T (odd)
        XNext n;
        n.ldsn = lds0 + (unsigned)((s & (XRING - 1)) * XGROUP + wave * XPPW * XFRAG);
        n.srd = (s & 1) ? vsrd : ksrd;
        n.so = s < ngroups ? (unsigned)(s >> 1) * XGROUP : 0x80000000u;      // zero-fill pieces behind the last group
        return n;
    };
    auto issue = [&](int s) {
        XNext n = next_of(s);
#pragma unroll
        for (int i = 0; i < XPPW; ++i) dma_piece(n.srd, voff, n.so + i * XFRAG, n.ldsn + i * XFRAG);
    };
    XNext nx;
    unsigned lane16 = lane * 16;
    auto group_sync = [&](int s) -> unsigned {
        wait_vmcnt<XPPW * (XRING - 2)>();
        __syncthreads();
        nx = next_of(s + XRING - 1);
        return lds0 + (unsigned)((s & (XRING - 1)) * XGROUP) + lane16;
    };
#pragma unroll
    for (int s = 0; s < XRING - 1; ++s) issue(s);
    // Q^T B fragments: k-step ks <- dims 16 ks + 8 hi .. + 7 of the lane's query (natural k order: K is packed to match)
    half8_t qf[32];
    {
        const int qrow = q0 + l32 < p.lq ? q0 + l32 : p.lq - 1;
        const half_t* qptr = p.q + ((long long)b * p.lq + qrow) * p.q_stride + 8 * hi;
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] = *(const half8_t*)(qptr + 16 * s);
    }
    static_for<256>([&](auto N) { acc_set<N>(0.f); });
    float m_run = -INFINITY, l_run = 0.f;
    int lane2;                                              // (fresh lane id behind the prologue: see the kernels of xattn_fused.hip)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
    const int hi2 = lane2 >> 5;
    lane16 = (unsigned)lane2 * 16;
#pragma unroll 1
    for (int t = 0; t < p.nt; ++t) {
        half8_t t0, t1, t2, t3, t4, t5;
        float16_t c0, c1;
        {
            const unsigned st = group_sync(2 * t);
            asm volatile(XG_SA : [c0] "=&v"(c0), [c1] "=&v"(c1), XTMP_OUT
                         : [st] "v"(st), [b0] "v"(qf[0]), [b1] "v"(qf[1]), [b2] "v"(qf[2]), [b3] "v"(qf[3]), [b4] "v"(qf[4]), [b5] "v"(qf[5]),
                           [b6] "v"(qf[6]), [b7] "v"(qf[7]), [b8] "v"(qf[8]), [b9] "v"(qf[9]), [b10] "v"(qf[10]), [b11] "v"(qf[11]),
                           [b12] "v"(qf[12]), [b13] "v"(qf[13]), [b14] "v"(qf[14]), [b15] "v"(qf[15]), XDMA_IN : "memory", "scc");
            asm volatile(XG_SB : [c0] "+v"(c0), [c1] "+v"(c1), XTMP_OUT
                         : [st] "v"(st), [b0] "v"(qf[16]), [b1] "v"(qf[17]), [b2] "v"(qf[18]), [b3] "v"(qf[19]), [b4] "v"(qf[20]), [b5] "v"(qf[21]),
                           [b6] "v"(qf[22]), [b7] "v"(qf[23]), [b8] "v"(qf[24]), [b9] "v"(qf[25]), [b10] "v"(qf[26]), [b11] "v"(qf[27]),
                           [b12] "v"(qf[28]), [b13] "v"(qf[29]), [b14] "v"(qf[30]), [b15] "v"(qf[31]), XDMA_IN : "memory", "scc");
        }
        // online softmax over the lane's keys 32 t + (r & 3) + 8 (r >> 2) + 4 hi (lane ^ 32 holds the others)
        const int key0 = t * 32 + 4 * hi2;
        float mx = -INFINITY;
        float sc[16];
        const bool whole = (t + 1) * 32 <= p.lk;             // wave-uniform: a tile without padding needs no mask
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = (c0[r] + c1[r]) * p.scale_log2;
            if (!whole) s = key0 + (r & 3) + 8 * (r >> 2) < p.lk ? s : -INFINITY;
            sc[r] = s; mx = fmaxf(mx, s);
        }
        mx = half_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sc[r] - m_new);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                         // exact deferred rescale (attn_kernel's rule); rare after the first tiles
            asm volatile("s_nop 15\ns_nop 15" ::: "memory"); // the previous tile's MFMAs must have written the accumulators
            static_for<256>([&](auto N) { acc_set<N>(acc_get<N>() * alpha); });
        }
        {
            const unsigned st = group_sync(2 * t + 1);
            asm volatile(XG_WD32 : XTMP_OUT : [st] "v"(st), [b0] "v"(pf[0]), [b1] "v"(pf[1]), XDMA_IN : "memory", "scc", XACC_CLOBBERS);
        }
    }
    const float inv = 1.0f / half_sum(l_run);
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    wait_vmcnt<0>();
    __syncthreads();                                        // every wave is done reading fragments: the ring is free
    // ---- store O (fp16 rows) row-coalesced through the wave's ring quarter: 8-B pieces in, whole rows out -------------------------------
    {
        typedef __attribute__((address_space(3))) uint2_t* lds_u2wptr_t;
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const int ln = lane_, l32e = lane_ & 31, hie = lane_ >> 5;
        const unsigned wbuf = lds0 + (unsigned)(wave * XGROUP);
        static_for<64>([&](auto JQ) {
            constexpr int j = JQ / 4, q = JQ % 4;
            const uint2_t h = {pack_h2f(acc_get<16 * j + 4 * q>() * inv, acc_get<16 * j + 4 * q + 1>() * inv),
                               pack_h2f(acc_get<16 * j + 4 * q + 2>() * inv, acc_get<16 * j + 4 * q + 3>() * inv)};
            const int pc8 = 8 * j + 2 * q + hie;            // 8-B piece of the 1-KiB row (channels 32 j + 8 q + 4 hi ..); 16-B block pc8 >> 1
            *(lds_u2wptr_t)(size_t)(wbuf + l32e * 1024 + ((((pc8 >> 1) ^ (l32e & 7)) << 4) | ((pc8 & 1) << 3))) = h;
        });
        asm volatile("" ::: "memory");
        half_t* const obase = p.o + ((long long)b * p.lq + q0) * p.o_stride + ln * 8;
#pragma unroll
        for (int kb = 0; kb < 32; kb += 8) {
            float4_t r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = lds_f4(wbuf + (kb + k) * 1024 + ((ln ^ ((kb + k) & 7)) << 4));
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (q0 + kb + k < p.lq) *(float4_t*)(obase + (long long)(kb + k) * p.o_stride) = r[k];
        }
    }
}

// K, V rows [bq * lk][stride] (fp16, 512 channels) -> fragment streams [bq][nt = ceil(lk / 32)][32 fragments][64 lanes][8 halves]:
// K: fragment ks of tile t = keys 32 t + l32, dims 16 ks + 8 hi + e;  V^T: fragment f = (channel tile 2 (f >> 2) + (f & 1), k-step
// (f >> 1) & 1) = channels 32 ct + l32, keys 32 t + 16 ks + 8 (e >> 2) + 4 hi + (e & 3) (the order the S^T accumulator hands P over in);
// keys >= lk are zero.
__global__ __launch_bounds__(256) void attn512x_pack_kernel(const half_t* __restrict__ k, long long k_stride, const half_t* __restrict__ v,
                                                            long long v_stride, int lk, int nt, long long total, half8_t* __restrict__ kp,
                                                            half8_t* __restrict__ vp) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;       // one 16-B unit of each stream per thread
    if (u >= total) return;
    const int lane = (int)(u & 63), f = (int)((u >> 6) & 31);
    const long long tt = u >> 11;                                        // b * nt + t
    const int t = (int)(tt % nt);
    const long long b = tt / nt;
    const int l32 = lane & 31, hi = lane >> 5;
    half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const int key = 32 * t + l32;
        kp[u] = key < lk ? *(const half8_t*)(k + (b * lk + key) * k_stride + 16 * f + 8 * hi) : z;
    }
    {
        const int ct = 2 * (f >> 2) + (f & 1), ks = (f >> 1) & 1;
        half8_t o = z;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = 32 * t + 16 * ks + 8 * (e >> 2) + 4 * hi + (e & 3);
            if (key < lk) o[e] = v[(b * lk + key) * v_stride + 32 * ct + l32];
        }
        vp[u] = o;
    }
}

}  // namespace

extern "C" int64_t uav_attention512_pack_bytes(int32_t bq, int32_t lk) {
    if (bq <= 0 || lk <= 0) return 0;
    return (int64_t)bq * ((lk + 31) / 32) * XGROUP;          // of EACH of the two streams
}

extern "C" int uav_attention512_pack_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, int32_t bq, int32_t lk, void* kp,
                                        void* vp, void* stream) {
    if (!k || !v || !kp || !vp) return UAV_EINVAL;
    if (bq <= 0 || lk <= 0 || (k_stride % 8) || (v_stride % 8) || k_stride < 512 || v_stride < 512) return UAV_ESHAPE;
    if (((size_t)k | (size_t)kp | (size_t)vp) & 15) return UAV_EALIGN;
    const int nt = (lk + 31) / 32;
    if ((long long)nt * XGROUP >= (1ll << 32)) return UAV_ESHAPE;       // one batch entry per buffer descriptor
    const long long total = (long long)bq * nt * 32 * 64;
    hipLaunchKernelGGL(attn512x_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)k,
                       (long long)k_stride, (const half_t*)v, (long long)v_stride, lk, nt, total, (half8_t*)kp, (half8_t*)vp);
    return uav_launch_status();
}

extern "C" int uav_attention512_packed_f16(const void* q, int64_t q_stride, const void* kp, const void* vp, void* o, int64_t o_stride,
                                           int32_t bq, int32_t lq, int32_t lk, float scale, void* stream) {
    if (!q || !kp || !vp || !o) return UAV_EINVAL;
    if (bq <= 0 || lq <= 0 || lk <= 0 || (q_stride % 8) || (o_stride % 8) || q_stride < 512 || o_stride < 512) return UAV_ESHAPE;
    if (((size_t)q | (size_t)o | (size_t)kp | (size_t)vp) & 15) return UAV_EALIGN;
    const int nt = (lk + 31) / 32;
    if ((long long)nt * XGROUP >= (1ll << 32)) return UAV_ESHAPE;
    Attn512xArgs a{(const half_t*)q, (long long)q_stride, (const char*)kp, (const char*)vp, (half_t*)o, (long long)o_stride, lq, lk, nt,
                   scale * 1.44269504088896341f};
    static UavDynLds lds;
    if (int rc = uav_set_dyn_lds(lds, (const void*)attn512x_kernel, XRING * XGROUP)) return rc;
    hipLaunchKernelGGL(attn512x_kernel, dim3((unsigned)((lq + 127) / 128), 1, (unsigned)bq), dim3(256), XRING * XGROUP, (hipStream_t)stream, a);
    return uav_launch_status();
}
