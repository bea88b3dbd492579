// The temporal attention sub-layer kernel template (NX = 0: alone; NX = 2: attn1 -> attn2 -> attn_temporal; FF = 1: ... -> ff), shared by
// tattn_fused.hip (instances <0,0>, <2,0>) and tattn_block_fused.hip (<2,1>): one translation unit per heavy instance keeps the build parallel.
#pragma once
#define UAV_HALF_REDUCE_PERMLANE 1
#include "xattn_common.h"

namespace {
// ---------------------------------------------------------------------------------------------------------------------------------
// Fused TEMPORAL attention sub-layer of BasicTransformerBlock (reference attention.py:555-560 `attn_temporal` step, TemporalAttention
// :626-733, RelativePositionBias :735-772, rotary-embedding-torch RoPE at :709-711) for the same 512-channel levels, T = 8 frames:
//
//     out = x + to_out( softmax( RoPE(to_q(n) * scale) . RoPE(to_k(n))^T + bias_h[tq][tk] ) . to_v(n) ) + b_out ,   n = LayerNorm(x)
//
// over the 8 tokens of one (batch, pixel).  Same skeleton as the kernel above — lane = token for the whole kernel, A fragments
// streamed through the LDS ring, named accumulators, row-coalesced stores — with these differences:
//   * a wave's 32 tokens are 4 neighbouring pixels x 8 frames (lane l32 = 4 t + px): rows (b T + t) hw + pix, i.e. 8 runs of 4 rows;
//   * three projections per head.  Q^T and K^T = W . Xn^T as above (lane = token); V = Xn . Wv^T with the MFMA operands the OTHER way
//     round (A = the token fragments in registers, B = the weight fragment): D[token][channel] has lane = CHANNEL and the tokens in
//     the registers — which is the A-operand layout of V^T in O^T = V^T . P^T, so no transpose exists anywhere;
//   * S^T[key][query] = K . Q^T is ONE 32 x 32 MFMA tile per head on register operands (K^T's D registers are K's A fragments, same
//     permuted k order); a query only sees the 8 keys of its own pixel: key row (r & 3) + 8 (r >> 2) + 4 hi has pixel r & 3 and frame
//     2 (r >> 2) + hi, so register r of lane l32 is live iff (r & 3) == (l32 & 3) — 4 keys in this lane, the other 4 in lane ^ 32; the
//     rest is masked to -inf (P = 0) and the PV MFMA runs over all 32 keys;
//   * roundings follow the three-launch chain: q, k, v rounded to fp16 where it stores the fused projection, RoPE in fp32 on the
//     scaled q / on k and rounded again, O rounded to fp16; P is rounded to fp16 here (the VALU kernel keeps it fp32).
constexpr int TGPH = 8;                        // groups per head: W_q, W_k, W_v, W_out (2 each)
constexpr int TNG = XHEADS * TGPH;
constexpr int TT = 8;                          // frames
constexpr int TTAB_REL = XTAB + 3 * XTABS;     // LDS behind the LayerNorm / bias tables of up to three sub-layers: relative-position bias [head][tq][hi][m] = bias[head][tq][2 m + hi] (2 KiB)
constexpr int TTAB_COS = TTAB_REL + 2048;      // RoPE cos [t][hi][2 q + pb] = cos[t][4 q + 2 hi + pb] (512 B), then sin
constexpr int TTAB_LN3 = TTAB_COS + 1024;      // gamma | beta of the LayerNorm BEHIND the sub-layer(s) (the block's norm3), when its output is asked for (4 KiB)
constexpr int TSMEM = TTAB_LN3 + 4096;

struct TattnArgs {
    const float* x; float* out; const float* gamma; const float* beta; const float* bias; float eps;
    const char* wq; const char* wk; const char* wv; const char* wo;
    const float* relbias; const float* rope_cos; const float* rope_sin;
    int n_batch; long long hw; float scale;
    XattnSub xs[2]; int lk; float xscale_log2;   // NX = 2: the block's two text cross-attention sub-layers in front (attn1, attn2)
    // optional: the NEXT LayerNorm of the block (norm3, in front of the feed-forward) applied to the rows this kernel writes, as fp16 operand
    // rows [M][512] — the rows are in the accumulators anyway, and the LayerNorm launch (4 B read + 2 B written per element) disappears
    half_t* ln_out; const float* ln_gamma; const float* ln_beta; float ln_eps;
    // FF = 1: the block's feed-forward sub-layer behind the three attention sub-layers (the whole BasicTransformerBlock in one launch)
    const float* ff_gamma; const float* ff_beta; const float* ff_down_bias; const float* ff_up_bias; const char* ff_w; float ff_eps;
    half_t* out_hilo;                            // FF = 1: the result as the fp16 hi | lo pair [M][1024] (instead of / beside out)
    // PI = 1: x is the INPUT of the Transformer3DModel's GroupNorm (fp32 rows); the kernel applies the GroupNorm (per-frame scale | shift
    // rows [B*T][512] from the statistics finalize), proj_in ('out'-packed fragments, bias) and goes on with the block on the result
    const float* gn_scale; const float* gn_shift; const char* w_in; const float* b_in;
};

// NX = 0: the temporal sub-layer alone.  NX = 2: attn1 -> attn2 -> attn_temporal of one BasicTransformerBlock (only_cross_attention) in ONE
// launch on the temporal tiling — a workgroup's 16 pixels x 8 frames lie inside one batch entry, which is all the cross-attention head
// loop asks of its 32 tokens —: the stream is read once and written once for three sub-layers, the rows between them stay in the
// accumulators and every LayerNorm but the first runs on them in registers.
// FF = 1 (with NX = 2): ... -> ff in the same launch: the whole block reads the stream once and writes it once (or only its hi | lo pair).
// PI = 1 (with NX = 2, FF = 1): GroupNorm apply -> proj_in in front: from the Transformer3DModel's GroupNorm input to the last block's output
// in one launch (attention.py:389-398 minus proj_out, whose residual + GroupNorm-statistics epilogue stays a conv launch).
constexpr int PIG = 16;                        // groups of proj_in: 8 k slices of 64 x (W_out-type groups 0, 1)
template <int NX, int FF, int PI = 0>
__global__ __launch_bounds__(256, 1) void tattn_sublayer_kernel(TattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned tiles_per_b = (unsigned)(p.hw / 16);
    const unsigned bb = __builtin_amdgcn_readfirstlane(blockIdx.x / tiles_per_b);
    const unsigned pt = blockIdx.x - bb * tiles_per_b;
    const long long rowbase = (long long)bb * TT * p.hw + (long long)pt * 16 + wave * 4;      // + t * hw + px
    const long long row = rowbase + (long long)(l32 >> 2) * p.hw + (l32 & 3);

    const unsigned voff = (unsigned)(wave * XPPW * XFRAG + lane * 16);
    constexpr int SG_X = PI * PIG;                         // first group of the first cross-attention sub-layer in the stream
    constexpr int SG_T = SG_X + NX * XNG;                  // first group of the temporal sub-layer
    auto next_of = [&](int s) -> XNext {
        XNext n;
        n.ldsn = lds0 + (unsigned)((s & (XRING - 1)) * XGROUP + wave * XPPW * XFRAG);
        if (PI && s < SG_X) { n.srd = make_srd(p.w_in, PIG * XGROUP); n.so = (unsigned)s * XGROUP; return n; }
        if (NX > 0 && s < SG_T) {                          // a cross-attention sub-layer: head r / 5, group j = r % 5 (0, 1: W_q; 2: K | V; 3, 4: W_out)
            const int sx = s - SG_X;
            const int u = sx >= XNG ? 1 : 0, r = sx - u * XNG;
            const int h = r / XGPH, j = r - h * XGPH;
            const XattnSub& S = p.xs[u];
            if (j < 2) { n.srd = make_srd(S.wq, XHEADS * 2 * XGROUP); n.so = (unsigned)((h * 2 + j) * XGROUP); }
            else if (j == 2) { n.srd = make_srd(S.kv + (long long)bb * XHEADS * XGROUP, XHEADS * XGROUP); n.so = (unsigned)(h * XGROUP); }
            else { n.srd = make_srd(S.wo, XHEADS * 2 * XGROUP); n.so = (unsigned)((h * 2 + (j - 3)) * XGROUP); }
            return n;
        }
        const int st = s - SG_T;                           // temporal: st = 8 h + j: j 0, 1: W_q; 2, 3: W_k; 4, 5: W_v; 6, 7: W_out
        if (FF && st >= TNG) {                             // the feed-forward's one linear stream (192 groups), then the zero-fill pieces
            const int sf = st - TNG;
            n.srd = make_srd(p.ff_w, FNG * XGROUP);
            n.so = sf < FNG ? (unsigned)sf * XGROUP : 0x80000000u;
            return n;
        }
        const int h = st >> 3, j = st & 7;
        const char* base = j < 2 ? p.wq : j < 4 ? p.wk : j < 6 ? p.wv : p.wo;
        n.srd = make_srd(base, XHEADS * 2 * XGROUP);
        n.so = (unsigned)((h * 2 + (j & 1)) * XGROUP);
        if (st >= TNG) n.so = 0x80000000u;                 // zero-fill pieces behind the last group (see the kernel above)
        return n;
    };
    auto issue = [&](int s) {
        XNext n = next_of(s);
#pragma unroll
        for (int i = 0; i < XPPW; ++i) dma_piece(n.srd, voff, n.so + i * XFRAG, n.ldsn + i * XFRAG);
    };
    XNext nx;
    unsigned lane16 = lane * 16;                            // (re-derived from a fresh lane id behind the prologue, see below)
    auto group_sync = [&](int s) -> unsigned {
        wait_vmcnt<XPPW * (XRING - 2)>();
        __syncthreads();
        nx = next_of(s + XRING - 1);
        return lds0 + (unsigned)((s & (XRING - 1)) * XGROUP) + lane16;
    };
#pragma unroll
    for (int s = 0; s < XRING - 1; ++s) issue(s);
    // ---- tables -> LDS ----------------------------------------------------------------------------------------------------------
    constexpr int TT_LN = XTAB + NX * XTABS;                // the temporal sub-layer's gamma | beta | bias behind the cross sub-layers'
    if (tid < 128) {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const unsigned tb = lds0 + XTAB + u * XTABS + tid * 16;
            *(lds_f4wptr_t)(size_t)tb = ((const float4_t*)p.xs[u].gamma)[tid];
            *(lds_f4wptr_t)(size_t)(tb + 2048) = ((const float4_t*)p.xs[u].beta)[tid];
            *(lds_f4wptr_t)(size_t)(tb + 4096) = ((const float4_t*)p.xs[u].bias)[tid];
        }
        *(lds_f4wptr_t)(size_t)(lds0 + TT_LN + tid * 16) = ((const float4_t*)p.gamma)[tid];
        *(lds_f4wptr_t)(size_t)(lds0 + TT_LN + 2048 + tid * 16) = ((const float4_t*)p.beta)[tid];
        *(lds_f4wptr_t)(size_t)(lds0 + TT_LN + 4096 + tid * 16) = ((const float4_t*)p.bias)[tid];
        if (!PI && p.ln_out) {
            *(lds_f4wptr_t)(size_t)(lds0 + TTAB_LN3 + tid * 16) = ((const float4_t*)p.ln_gamma)[tid];
            *(lds_f4wptr_t)(size_t)(lds0 + TTAB_LN3 + 2048 + tid * 16) = ((const float4_t*)p.ln_beta)[tid];
        }
        if (PI) *(lds_f4wptr_t)(size_t)(lds0 + TTAB_LN3 + tid * 16) = ((const float4_t*)p.b_in)[tid];     // proj_in bias where norm3's tables would be
    } else {
        typedef __attribute__((address_space(3))) float* lds_fptr_t;
        const int u = tid - 128;                            // 128 threads: 512 bias entries (4 each), 128 cos + 128 sin (1 + 1 each)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = u * 4 + k;                        // e = ((h * 8 + tq) * 2 + hi_) * 4 + m
            const int m = e & 3, hi_ = (e >> 2) & 1, tq = (e >> 3) & 7, h = e >> 6;
            *(lds_fptr_t)(size_t)(lds0 + TTAB_REL + e * 4) = p.relbias[(h * TT + tq) * TT + 2 * m + hi_];
        }
        {
            const int pb = u & 1, q = (u >> 1) & 3, hi_ = (u >> 3) & 1, t = u >> 4;      // u = ((t * 2 + hi_) * 4 + q) * 2 + pb
            const int pair = 4 * q + 2 * hi_ + pb;
            *(lds_fptr_t)(size_t)(lds0 + TTAB_COS + u * 4) = p.rope_cos[t * 16 + pair];
            *(lds_fptr_t)(size_t)(lds0 + TTAB_COS + 512 + u * 4) = p.rope_sin[t * 16 + pair];
        }
    }
    half8_t xn[32];
    if constexpr (PI) {
        // ---- GroupNorm apply + proj_in: the scale | shift rows of the workgroup's 8 frames (32 KiB) sit in ring slot 3, which the stream
        // does not touch before the first group is walked (its pieces go out between the MFMAs of group 0, behind a barrier every wave
        // reaches with its rows normalised); ONE read of x (the statistics are the GroupNorm's, already finalized) ---------------------------
        {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                   // 2048 16-B units: frame f, part (scale / shift), channel quad c4
                const int idx = k * 256 + tid, f = idx >> 8, part = (idx >> 7) & 1, c4 = idx & 127;
                const float* src = (part ? p.gn_shift : p.gn_scale) + ((long long)(bb * TT + f) * XC + c4 * 4);
                *(lds_f4wptr_t)(size_t)(lds0 + 3 * XGROUP + f * 4096 + part * 2048 + c4 * 16) = *(const float4_t*)src;
            }
        }
        __syncthreads();                                    // tables visible
        const float* xr = p.x + row * XC + 4 * hi;
        const unsigned gt = lds0 + 3 * XGROUP + (l32 >> 2) * 4096 + 16 * hi;
        static_for<16>([&](auto J) {
            constexpr int j = J;
            static_for<4>([&](auto Q) {
                constexpr int q = Q;
                const float4_t v = *(const float4_t*)(xr + 32 * j + 8 * q);
                const float4_t sc = lds_f4(gt + (32 * j + 8 * q) * 4), sh = lds_f4(gt + 2048 + (32 * j + 8 * q) * 4);
                const float4_t bi = lds_f4(lds0 + TTAB_LN3 + (32 * j + 8 * q + 4 * hi) * 4);
                static_for<4>([&](auto I) {
                    constexpr int i = I;
                    xn[2 * j + (q >> 1)][4 * (q & 1) + i] = (half_t)(v[i] * sc[i] + sh[i]);      // gn_apply_kernel's arithmetic and rounding
                    acc_set<16 * j + 4 * q + i>(bi[i]);
                });
            });
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        });
        // tok^T [512 ch][32 tokens] = W_in . N^T + b_in: 8 k slices of 64 = 8 x (W_out-type groups 0, 1) on the named accumulators
        {
            half8_t t0, t1, t2, t3, t4, t5;
            static_for<8>([&](auto Hh) {
                constexpr int h = Hh;
                {
                    const unsigned st = group_sync(2 * h);
                    asm volatile(XG_WO0 : XTMP_OUT : [st] "v"(st), [b0] "v"(xn[4 * h]), [b1] "v"(xn[4 * h + 1]), [b2] "v"(xn[4 * h + 2]),
                                 [b3] "v"(xn[4 * h + 3]), XDMA_IN : "memory", "scc", XACC_CLOBBERS);
                }
                {
                    const unsigned st = group_sync(2 * h + 1);
                    asm volatile(XG_WO1 : XTMP_OUT : [st] "v"(st), [b0] "v"(xn[4 * h]), [b1] "v"(xn[4 * h + 1]), [b2] "v"(xn[4 * h + 2]),
                                 [b3] "v"(xn[4 * h + 3]), XDMA_IN : "memory", "scc", XACC_CLOBBERS);
                }
            });
        }
        // norm1 on the token rows the accumulators hold (+ attn1's output bias): from here on as if they had been read from HBM
        mid_layernorm(xn, lds0 + XTAB, p.xs[0].eps, hi);
    } else {
        // ---- LayerNorm statistics (first read), operand fragments + accumulators (second read): as in the kernel above -------------------
        const float* xr = p.x + row * XC + 4 * hi;
        const float c0 = p.x[row * XC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jb = 0; jb < 16; jb += 4) {
#pragma unroll
            for (int j = jb; j < jb + 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4_t v = *(const float4_t*)(xr + 32 * j + 8 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = v[i] - c0; s1 += d; s2 += d * d; }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        s1 = half_sum(s1); s2 = half_sum(s2);
        const float m1 = s1 * (1.0f / XC);
        const float mean = c0 + m1;
        const float rstd = rsqrtf(fmaxf(s2 * (1.0f / XC) - m1 * m1, 0.f) + (NX > 0 ? p.xs[0].eps : p.eps));      // (the tables at XTAB are the first sub-layer's)
        __syncthreads();                                        // tables visible
        static_for<16>([&](auto J) {
            constexpr int j = J;
            static_for<4>([&](auto Q) {
                constexpr int q = Q;
                const float4_t v = *(const float4_t*)(xr + 32 * j + 8 * q);
                const unsigned ta = lds0 + XTAB + (32 * j + 8 * q + 4 * hi) * 4;
                const float4_t g = lds_f4(ta), be = lds_f4(ta + 2048), bo = lds_f4(ta + 4096);
                static_for<4>([&](auto I) {
                    constexpr int i = I;
                    xn[2 * j + (q >> 1)][4 * (q & 1) + i] = (half_t)((v[i] - mean) * rstd * g[i] + be[i]);
                    acc_set<16 * j + 4 * q + i>(v[i] + bo[i]);
                });
            });
            if (j & 1) __builtin_amdgcn_sched_barrier(0);       // batches of 8 loads (this prologue carries the row arithmetic of the frame-strided tile on top)
        });
    }

    auto temporal_heads = [&](half8_t (&xn)[32]) {
    // (the lane's pixel / frame from a fresh lane id: kept live from the row computation at the top they were spilled across the prologue)
    int lane2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
    const int px = lane2 & 3, tq = (lane2 & 31) >> 2, hi = lane2 >> 5;
    lane16 = (unsigned)lane2 * 16;
    // ---- heads ------------------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int h = 0; h < XHEADS; ++h) {
        half8_t t0, t1, t2, t3, t4, t5;
        const int sg = SG_T + h * TGPH;
        const unsigned tc = lds0 + TTAB_COS + ((tq * 2 + hi) * 8) * 4;
        // RoPE on the lane's first 32 head channels (tile 0: registers r <-> channel (r & 3) + 8 (r >> 2) + 4 hi; pairs (4 q, 4 q + 1),
        // (4 q + 2, 4 q + 3) of the registers are channel pairs (2 i, 2 i + 1), angle index 4 q + 2 hi + pb)
        auto rope16 = [&](const float (&a)[16], half8_t& f0, half8_t& f1) {
            const float4_t ca = lds_f4(tc), cb = lds_f4(tc + 16), sa = lds_f4(tc + 512), sb = lds_f4(tc + 528);
            const float cs[8] = {ca[0], ca[1], ca[2], ca[3], cb[0], cb[1], cb[2], cb[3]};
            const float sn[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {                // pair pr = 2 q + pb <-> registers 4 q + 2 pb, 4 q + 2 pb + 1
                const int r = 4 * (pr >> 1) + 2 * (pr & 1);
                const float u = a[r], w = a[r + 1];
                const half_t e0 = (half_t)(u * cs[pr] - w * sn[pr]), e1 = (half_t)(w * cs[pr] + u * sn[pr]);
                if (r < 8) { f0[r] = e0; f0[r + 1] = e1; } else { f1[r - 8] = e0; f1[r - 7] = e1; }
            }
        };
        // ---- Q^T = Wq_h . Xn^T -> fp16 (as stored by the chain) -> * scale -> RoPE -> fp16 B fragments ------------------------------
        half8_t qf[4];
        {
            float16_t q0, q1;
            {
                const unsigned st = group_sync(sg);
                asm volatile(XG_WQ_FIRST : [q0] "=&v"(q0), [q1] "=&v"(q1), XTMP_OUT
                             : [st] "v"(st), [b0] "v"(xn[0]), [b1] "v"(xn[1]), [b2] "v"(xn[2]), [b3] "v"(xn[3]), [b4] "v"(xn[4]), [b5] "v"(xn[5]),
                               [b6] "v"(xn[6]), [b7] "v"(xn[7]), [b8] "v"(xn[8]), [b9] "v"(xn[9]), [b10] "v"(xn[10]), [b11] "v"(xn[11]),
                               [b12] "v"(xn[12]), [b13] "v"(xn[13]), [b14] "v"(xn[14]), [b15] "v"(xn[15]), XDMA_IN : "memory", "scc");
            }
            {
                const unsigned st = group_sync(sg + 1);
                asm volatile(XG_WQ : [q0] "+v"(q0), [q1] "+v"(q1), XTMP_OUT
                             : [st] "v"(st), [b0] "v"(xn[16]), [b1] "v"(xn[17]), [b2] "v"(xn[18]), [b3] "v"(xn[19]), [b4] "v"(xn[20]), [b5] "v"(xn[21]),
                               [b6] "v"(xn[22]), [b7] "v"(xn[23]), [b8] "v"(xn[24]), [b9] "v"(xn[25]), [b10] "v"(xn[26]), [b11] "v"(xn[27]),
                               [b12] "v"(xn[28]), [b13] "v"(xn[29]), [b14] "v"(xn[30]), [b15] "v"(xn[31]), XDMA_IN : "memory", "scc");
            }
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = (float)(half_t)q0[r] * p.scale;
            rope16(a, qf[0], qf[1]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { qf[2][e] = (half_t)((float)(half_t)q1[e] * p.scale); qf[3][e] = (half_t)((float)(half_t)q1[8 + e] * p.scale); }
        }
        // ---- K^T = Wk_h . Xn^T -> fp16 -> RoPE -> fp16: its D registers are the A fragments of K in S^T = K . Q^T ----------------------
        half8_t kf[4];
        {
            float16_t q0, q1;
            {
                const unsigned st = group_sync(sg + 2);
                asm volatile(XG_WQ_FIRST : [q0] "=&v"(q0), [q1] "=&v"(q1), XTMP_OUT
                             : [st] "v"(st), [b0] "v"(xn[0]), [b1] "v"(xn[1]), [b2] "v"(xn[2]), [b3] "v"(xn[3]), [b4] "v"(xn[4]), [b5] "v"(xn[5]),
                               [b6] "v"(xn[6]), [b7] "v"(xn[7]), [b8] "v"(xn[8]), [b9] "v"(xn[9]), [b10] "v"(xn[10]), [b11] "v"(xn[11]),
                               [b12] "v"(xn[12]), [b13] "v"(xn[13]), [b14] "v"(xn[14]), [b15] "v"(xn[15]), XDMA_IN : "memory", "scc");
            }
            {
                const unsigned st = group_sync(sg + 3);
                asm volatile(XG_WQ : [q0] "+v"(q0), [q1] "+v"(q1), XTMP_OUT
                             : [st] "v"(st), [b0] "v"(xn[16]), [b1] "v"(xn[17]), [b2] "v"(xn[18]), [b3] "v"(xn[19]), [b4] "v"(xn[20]), [b5] "v"(xn[21]),
                               [b6] "v"(xn[22]), [b7] "v"(xn[23]), [b8] "v"(xn[24]), [b9] "v"(xn[25]), [b10] "v"(xn[26]), [b11] "v"(xn[27]),
                               [b12] "v"(xn[28]), [b13] "v"(xn[29]), [b14] "v"(xn[30]), [b15] "v"(xn[31]), XDMA_IN : "memory", "scc");
            }
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = (float)(half_t)q0[r];
            rope16(a, kf[0], kf[1]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { kf[2][e] = (half_t)q1[e]; kf[3][e] = (half_t)q1[8 + e]; }
        }
        // ---- S^T [32 keys][32 queries] on register operands; softmax over the 8 keys of the query's own pixel ------------------------
        // (asm with VGPR results: left to hipcc the MFMA intrinsic takes its result registers from the accumulator file — a[0:15], i.e. the
        //  NAMED accumulator tile 0 of this kernel, which the compiler cannot know is live; the build audit caught exactly that)
        float16_t sacc;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %5, 0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %2, %6, %0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %4, %8, %0\n" XNOP
                     : "=&v"(sacc) : "v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));
        const float4_t rb = lds_f4(lds0 + TTAB_REL + (((h * 8 + tq) * 2 + hi) * 4) * 4);        // bias[h][tq][2 m + hi], m = 0 .. 3
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = ((r & 3) == px) ? sacc[r] + rb[r >> 2] : -INFINITY;
            sacc[r] = s; mx = fmaxf(mx, s);
        }
        mx = half_max(mx);
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f((sacc[r] - mx) * 1.44269504088896341f);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        ps = half_sum(ps);
        const float inv = 1.0f / ps;
        // ---- V = Xn . Wv_h^T (lane = channel, registers = tokens) -> fp16 = the A fragments of V^T ------------------------------------
        half8_t of[4];
        {
            float16_t q0, q1;
            {
                const unsigned st = group_sync(sg + 4);
                asm volatile(XG_WV_FIRST : [q0] "=&v"(q0), [q1] "=&v"(q1), XTMP_OUT
                             : [st] "v"(st), [b0] "v"(xn[0]), [b1] "v"(xn[1]), [b2] "v"(xn[2]), [b3] "v"(xn[3]), [b4] "v"(xn[4]), [b5] "v"(xn[5]),
                               [b6] "v"(xn[6]), [b7] "v"(xn[7]), [b8] "v"(xn[8]), [b9] "v"(xn[9]), [b10] "v"(xn[10]), [b11] "v"(xn[11]),
                               [b12] "v"(xn[12]), [b13] "v"(xn[13]), [b14] "v"(xn[14]), [b15] "v"(xn[15]), XDMA_IN : "memory", "scc");
            }
            {
                const unsigned st = group_sync(sg + 5);
                asm volatile(XG_WV : [q0] "+v"(q0), [q1] "+v"(q1), XTMP_OUT
                             : [st] "v"(st), [b0] "v"(xn[16]), [b1] "v"(xn[17]), [b2] "v"(xn[18]), [b3] "v"(xn[19]), [b4] "v"(xn[20]), [b5] "v"(xn[21]),
                               [b6] "v"(xn[22]), [b7] "v"(xn[23]), [b8] "v"(xn[24]), [b9] "v"(xn[25]), [b10] "v"(xn[26]), [b11] "v"(xn[27]),
                               [b12] "v"(xn[28]), [b13] "v"(xn[29]), [b14] "v"(xn[30]), [b15] "v"(xn[31]), XDMA_IN : "memory", "scc");
            }
            half8_t vf[2][2];
#pragma unroll
            for (int e = 0; e < 8; ++e) { vf[0][0][e] = (half_t)q0[e]; vf[0][1][e] = (half_t)q0[8 + e]; vf[1][0][e] = (half_t)q1[e]; vf[1][1][e] = (half_t)q1[8 + e]; }
            // O^T [64 ch][32 queries] = V^T . P^T
            float16_t o0, o1;
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %6, 0\n"
                         "v_mfma_f32_32x32x16_f16 %1, %4, %6, 0\n"
                         "v_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n"
                         "v_mfma_f32_32x32x16_f16 %1, %5, %7, %1\n" XNOP
                         : "=&v"(o0), "=&v"(o1) : "v"(vf[0][0]), "v"(vf[0][1]), "v"(vf[1][0]), "v"(vf[1][1]), "v"(pf[0]), "v"(pf[1]));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                of[0][e] = (half_t)(o0[e] * inv); of[1][e] = (half_t)(o0[8 + e] * inv);
                of[2][e] = (half_t)(o1[e] * inv); of[3][e] = (half_t)(o1[8 + e] * inv);
            }
        }
        // ---- acc += Wout[:, head h] . O^T ---------------------------------------------------------------------------------------------
        {
            const unsigned st = group_sync(sg + 6);
            asm volatile(XG_WO0 : XTMP_OUT : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), [b2] "v"(of[2]), [b3] "v"(of[3]), XDMA_IN
                         : "memory", "scc", XACC_CLOBBERS);
        }
        {
            const unsigned st = group_sync(sg + 7);
            asm volatile(XG_WO1 : XTMP_OUT : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), [b2] "v"(of[2]), [b3] "v"(of[3]), XDMA_IN
                         : "memory", "scc", XACC_CLOBBERS);
        }
    }
    };
    if constexpr (NX > 0) {
        // ---- attn1, attn2 (text cross-attention) on the same tile, then the temporal sub-layer's LayerNorm on their output ---------------
        unsigned long long ts_[12];
        xattn_heads<0>(SG_X, xn, group_sync, nx, voff, hi, p.lk, p.xscale_log2, ts_, false);
        mid_layernorm(xn, lds0 + XTAB + XTABS, p.xs[1].eps, hi);
        xattn_heads<0>(SG_X + XNG, xn, group_sync, nx, voff, hi, p.lk, p.xscale_log2, ts_, false);
        // (a second fragment array: hipcc gives the temporal loop's fragments other registers than the cross loops', and moving one set
        //  onto the other through the full register file went through scratch — 33 spilled fragments)
        half8_t xt[32];
        mid_layernorm(xt, lds0 + TT_LN, p.eps, hi);
        temporal_heads(xt);
        if constexpr (FF) {
            // ---- the feed-forward on the same tile: its tables (gamma | beta | b_down, then b_up: 22 KiB) go where the attention
            // sub-layers' tables were — every wave is past its last use of them behind this barrier.  (Plain loads: the wait the compiler
            // puts in front of the LDS writes also drains the DMA pieces in flight — they are older —, the count group_sync waits for
            // stays an upper bound.)
            asm volatile("s_nop 15\ns_nop 15" ::: "memory");
            __syncthreads();
            {
                int lf;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lf));
                const int t2 = wave * 64 + lf;
                if (t2 < 128) {
                    *(lds_f4wptr_t)(size_t)(lds0 + XTAB + t2 * 16) = ((const float4_t*)p.ff_gamma)[t2];
                    *(lds_f4wptr_t)(size_t)(lds0 + XTAB + 2048 + t2 * 16) = ((const float4_t*)p.ff_beta)[t2];
                    *(lds_f4wptr_t)(size_t)(lds0 + XTAB + 4096 + t2 * 16) = ((const float4_t*)p.ff_down_bias)[t2];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *(lds_f4wptr_t)(size_t)(lds0 + XTAB + XTABS + (k * 256 + t2) * 16) = ((const float4_t*)p.ff_up_bias)[k * 256 + t2];
            }
            __syncthreads();
            half8_t xf[32];
            int lf2;                                        // (the lane's half from a fresh lane id: the kernel-top `hi` kept live across the temporal heads
                                                            //  for this one use was parked in a0 / a1 — the NAMED accumulators — by hipcc; the build audit caught it)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lf2));
            mid_layernorm(xf, lds0 + XTAB, p.ff_eps, lf2 >> 5);
            ff_slices(SG_T + TNG, xf, group_sync, nx, voff, lane16, lds0 + XTAB + XTABS);
        }
    }
    if constexpr (NX == 0) temporal_heads(xn);
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    wait_vmcnt<0>();
    __syncthreads();
    // ---- store: row-coalesced through the idle ring (rows of the wave in lane order: 4 t + px) -----------------------------------------
    {
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const int ln = lane_, l32e = lane_ & 31, hie = lane_ >> 5;
        const unsigned wbuf = (unsigned)(size_t)(lptr_t)smem + (unsigned)(wave * XGROUP);
        float* const obase = p.out + rowbase * XC + ln * 4;
        if (p.ln_out) {
            // ---- the block's next LayerNorm on the finished rows (two passes over the accumulators like layernorm_kernel), fp16 rows out: the
            // lane's 8-B pieces (4 channels) into the wave's ring quarter — 32 rows x 1 KiB, 16-B block pb of row r at pb ^ (r & 7) —, whole rows back
            typedef __attribute__((address_space(3))) uint2_t* lds_u2wptr_t;
            float sm = 0.f;
            static_for<256>([&](auto N) { sm += acc_get<N>(); });
            sm = half_sum(sm);
            const float mean3 = sm * (1.0f / XC);
            float sq = 0.f;
            static_for<256>([&](auto N) { const float d = acc_get<N>() - mean3; sq += d * d; });
            sq = half_sum(sq);
            const float rstd3 = rsqrtf(sq * (1.0f / XC) + p.ln_eps);
            static_for<64>([&](auto JQ) {
                constexpr int j = JQ / 4, q = JQ % 4;
                const unsigned ta = (unsigned)(size_t)(lptr_t)smem + TTAB_LN3 + (32 * j + 8 * q + 4 * hie) * 4;
                const float4_t g = lds_f4(ta), be = lds_f4(ta + 2048);
                const float v0 = acc_get<16 * j + 4 * q>(), v1 = acc_get<16 * j + 4 * q + 1>(), v2 = acc_get<16 * j + 4 * q + 2>(), v3 = acc_get<16 * j + 4 * q + 3>();
                const uint2_t h = {pack_h2f((v0 - mean3) * rstd3 * g[0] + be[0], (v1 - mean3) * rstd3 * g[1] + be[1]),
                                   pack_h2f((v2 - mean3) * rstd3 * g[2] + be[2], (v3 - mean3) * rstd3 * g[3] + be[3])};
                const int pc8 = 8 * j + 2 * q + hie;        // 8-B piece of the 1-KiB row; 16-B block pc8 >> 1
                *(lds_u2wptr_t)(size_t)(wbuf + l32e * 1024 + ((((pc8 >> 1) ^ (l32e & 7)) << 4) | ((pc8 & 1) << 3))) = h;
            });
            asm volatile("" ::: "memory");
            half_t* const nbase = p.ln_out + rowbase * XC + ln * 8;
#pragma unroll
            for (int kb = 0; kb < 32; kb += 8) {
                float4_t r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = lds_f4(wbuf + (kb + k) * 1024 + ((ln ^ ((kb + k) & 7)) << 4));
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    *(float4_t*)(nbase + ((long long)((kb + k) >> 2) * p.hw + ((kb + k) & 3)) * XC) = r[k];
            }
            asm volatile("" ::: "memory");
        }
        if constexpr (FF) {
            if (p.out_hilo) {                               // the result as the hi | lo operand pair of proj_out (see ff_sublayer_kernel), frame-strided rows
                typedef __attribute__((address_space(3))) uint2_t* lds_u2wptr_t;
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    static_for<64>([&](auto JQ) {
                        constexpr int j = JQ / 4, q = JQ % 4;
                        float v[4] = {acc_get<16 * j + 4 * q>(), acc_get<16 * j + 4 * q + 1>(), acc_get<16 * j + 4 * q + 2>(), acc_get<16 * j + 4 * q + 3>()};
                        if (part) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = v[i] - (float)(half_t)v[i];
                        }
                        const uint2_t h = {pack_h2f(v[0], v[1]), pack_h2f(v[2], v[3])};
                        const int pc8 = 8 * j + 2 * q + hie;
                        *(lds_u2wptr_t)(size_t)(wbuf + l32e * 1024 + ((((pc8 >> 1) ^ (l32e & 7)) << 4) | ((pc8 & 1) << 3))) = h;
                    });
                    asm volatile("" ::: "memory");
                    half_t* const nbase = p.out_hilo + rowbase * (2 * XC) + part * XC + ln * 8;
#pragma unroll
                    for (int kb = 0; kb < 32; kb += 8) {
                        float4_t r[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) r[k] = lds_f4(wbuf + (kb + k) * 1024 + ((ln ^ ((kb + k) & 7)) << 4));
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            *(float4_t*)(nbase + ((long long)((kb + k) >> 2) * p.hw + ((kb + k) & 3)) * (2 * XC)) = r[k];
                    }
                    asm volatile("" ::: "memory");
                }
            }
        }
        if (!FF || p.out) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            static_for<32>([&](auto JQ) {
                constexpr int j = JQ / 4, q = JQ % 4;
                float4_t v;
                if (hh == 0) v = float4_t{acc_get<16 * j + 4 * q>(), acc_get<16 * j + 4 * q + 1>(), acc_get<16 * j + 4 * q + 2>(), acc_get<16 * j + 4 * q + 3>()};
                else v = float4_t{acc_get<128 + 16 * j + 4 * q>(), acc_get<128 + 16 * j + 4 * q + 1>(), acc_get<128 + 16 * j + 4 * q + 2>(), acc_get<128 + 16 * j + 4 * q + 3>()};
                const int pc = 8 * j + 2 * q + hie;
                *(lds_f4wptr_t)(size_t)(wbuf + l32e * 1024 + ((pc ^ (l32e & 7)) << 4)) = v;
            });
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 32; kb += 8) {
                float4_t r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = lds_f4(wbuf + (kb + k) * 1024 + ((ln ^ ((kb + k) & 7)) << 4));
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    *(float4_t*)(obase + ((long long)((kb + k) >> 2) * p.hw + ((kb + k) & 3)) * XC + hh * 256) = r[k];
            }
            asm volatile("" ::: "memory");
        }
        }
    }
}

}  // namespace

// launches of the instances, one definition per translation unit (UavTattnLaunch: the kernel arguments are an anonymous-namespace type, so they
// cross the TU boundary as an opaque pointer to a struct both sides compile from this header)
int uav_tattn_run_attn(const void* tattn_args, int nx, dim3 grid, hipStream_t stream);       // tattn_fused.hip: <0,0>, <2,0>
int uav_tattn_run_block_ff(const void* tattn_args, dim3 grid, hipStream_t stream);           // tattn_block_fused.hip: <2,1>
int uav_tattn_run_block_pi(const void* tattn_args, dim3 grid, hipStream_t stream);           // tattn_block_pi_fused.hip: <2,1,1>
