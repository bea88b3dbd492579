// Fused text cross-attention sub-layer of BasicTransformerBlock (reference models_video/attention.py:523-564, steps
// `attn1` with only_cross_attention / `attn2`; CrossAttention.forward :177-238) for the 512-channel levels of the UNet
// (8 heads x 64, 77 text keys):
//
//     out = x + to_out( softmax( to_q(LayerNorm(x)) . K^T * scale ) . V ) + b_out
//
// in ONE kernel that reads the fp32 token stream once and writes it once (8 B per element instead of the 24 B of the four
// launches it replaces: LayerNorm 4 + 2, to_q 2 + 2, attention 2 + 2, to_out 2 + 4 + 4).
//
// Layout.  A workgroup of four waves (one per SIMD, 512 registers each) owns 128 tokens, a wave 32 of them, LANE = TOKEN for the
// whole kernel: every GEMM is computed transposed (D^T = W . X^T, the "swapped" MFMA of the conv kernels), so the D layout of one
// product — lane (token, half h), register r = channel (r & 3) + 8 (r >> 2) + 4 h of a 32-channel tile — IS the B-operand layout of
// the next one once the k index is read in the order  slot e of half h  <->  k = 16 ks + 8 (e >> 2) + 4 h + (e & 3)  (a contraction
// index may be permuted freely as long as both operands agree).  All A operands — W_q, K, V^T, W_out — are therefore PRE-PACKED on
// that k order into 1-KiB MFMA fragments (64 lanes x 16 B, exactly what one `ds_read_b128` hands a wave) and streamed in
// consumption order through an LDS ring by LDS-DMA (`buffer_load ... lds`, contiguous 32-KiB groups straight from L2, the pieces of
// the group three ahead issued BETWEEN the MFMAs of the current one: in a burst behind the barrier they cost ~80 cycles each with the
// matrix pipe idle — 0.98 ms per launch at M = 409 600 in the first version, run 1 of round 6); nothing an
// accumulator holds ever moves between lanes except the two half-wave reductions of LayerNorm and softmax:
//
//   x (fp32, D layout) -> 256 accumulators (the residual is the accumulators' initial value) -> LayerNorm in registers ->
//   Xn fp16 B fragments (128 VGPRs) -> per head: Q^T = Wq_h Xn^T (64 MFMA) -> fp16 -> S^T = K_h Q^T (12) -> softmax over the
//   lane's keys -> P fp16 -> O^T = V_h^T P^T (12) -> fp16 -> acc += Wout[:, h] O^T (64) -> store.
//
// A group (32 fragments) is walked by ONE asm statement: six `ds_read_b128` in flight, every MFMA waits for exactly its
// fragment (`lgkmcnt`) and the register it frees is refilled at once — left to itself hipcc issues read, wait(0), MFMA with a
// single fragment register (measured on the first version of this file: the 128 VGPRs of Xn leave it no room to do better).
//
// Roundings are those of the unfused chain (LayerNorm output, Q, P, O rounded to fp16; everything else fp32), so the result
// agrees with it to fp32 summation order (tests/test_kernels_gpu.py::test_fused_cross_attention_sublayer).
#include "xattn_common.h"

namespace {

// TR = 1: development instance that stamps s_memtime at the phase boundaries (tools/trace_xattn.py); the product is TR = 0
template <int TR>
__global__ __launch_bounds__(256, 1) void xattn_sublayer_kernel(XattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (TR) ts[0] = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const long long tile0 = (long long)blockIdx.x * 128;
    const int b = __builtin_amdgcn_readfirstlane((int)(tile0 / p.rows_per_kv));
    const long long row = tile0 + wave * 32 + l32;

    // ---- the stream of A fragments: 40 groups of 32 KiB per sub-layer, eight 1-KiB pieces per wave and group; a second sub-layer's
    // groups follow the first's without a gap (the ring never drains between the two) ---------------------------------------------
    const unsigned voff = (unsigned)(wave * XPPW * XFRAG + lane * 16);
    const int ngroups = __builtin_amdgcn_readfirstlane(p.nsub * XNG);
    auto next_of = [&](int s) -> XNext {                    // s: group index over all sub-layers; inside one: head s / 5, group j = s % 5 (0, 1: W_q; 2: K | V; 3, 4: W_out)
        XNext n;
        n.ldsn = lds0 + (unsigned)((s & (XRING - 1)) * XGROUP + wave * XPPW * XFRAG);
        const int u = s >= XNG ? 1 : 0, r = s - u * XNG;
        const int h = r / XGPH, j = r - h * XGPH;
        const XattnSub& S = p.sub[u && p.nsub > 1 ? 1 : 0];
        if (j < 2) { n.srd = make_srd(S.wq, XHEADS * 2 * XGROUP); n.so = (unsigned)((h * 2 + j) * XGROUP); }
        else if (j == 2) { n.srd = make_srd(S.kv + (long long)b * XHEADS * XGROUP, XHEADS * XGROUP); n.so = (unsigned)(h * XGROUP); }
        else { n.srd = make_srd(S.wo, XHEADS * 2 * XGROUP); n.so = (unsigned)((h * 2 + (j - 3)) * XGROUP); }
        // behind the last group of the tile: the same eight pieces with every lane out of the descriptor's range — the hardware fetches
        // nothing and zero-fills a slot nobody reads again, and the vmcnt arithmetic below stays the same for every group
        if (s >= ngroups) n.so = 0x80000000u;
        return n;
    };
    auto issue = [&](int s) {                              // prologue: a whole group at once
        XNext n = next_of(s);
#pragma unroll
        for (int i = 0; i < XPPW; ++i) dma_piece(n.srd, voff, n.so + i * XFRAG, n.ldsn + i * XFRAG);
    };
    // before group s is read: this wave's pieces of it have landed (the two groups issued behind it may still fly: 16 pieces),
    // then every wave's have (barrier) — which also says every wave is done with the group before it, whose slot the group three
    // ahead is written into WHILE this group is multiplied (the pieces sit between the MFMAs of the asm walk).
    XNext nx;
    auto group_sync = [&](int s) -> unsigned {
        wait_vmcnt<XPPW * (XRING - 2)>();
        __syncthreads();
        nx = next_of(s + XRING - 1);
        return lds0 + (unsigned)((s & (XRING - 1)) * XGROUP) + lane * 16;
    };

#pragma unroll
    for (int s = 0; s < XRING - 1; ++s) issue(s);
    // ---- tables -> LDS; LayerNorm statistics of the lane's token ------------------------------------------------------------------
    {
        const int u = tid >> 7, t7 = tid & 127;             // threads 0 .. 127: first sub-layer, 128 .. 255: second
        if (u < p.nsub) {
            const XattnSub& S = p.sub[u];
            const unsigned tb = lds0 + XTAB + u * XTABS + t7 * 16;
            *(lds_f4wptr_t)(size_t)tb = ((const float4_t*)S.gamma)[t7];
            *(lds_f4wptr_t)(size_t)(tb + 2048) = ((const float4_t*)S.beta)[t7];
            *(lds_f4wptr_t)(size_t)(tb + 4096) = ((const float4_t*)S.bias)[t7];
        }
    }
    // The lane holds half of its token's row (channels 32 j + 8 q + 4 hi + i), lane ^ 32 the other half.  The row is read TWICE —
    // once for the statistics, once (from L2) for the operand and the residual — because 256 fp32 values + the 128 operand registers
    // they turn into do not fit beside each other in the 256 architectural VGPRs (the accumulator file cannot feed the VALU; hipcc
    // spilled 118 ... 565 registers per lane on every single-read form tried).  Statistics in one pass on values shifted by the
    // row's first element c: mean = c + E[x - c], var = E[(x - c)^2] - E[x - c]^2 (no cancellation: |mean - c| is of the order of
    // the spread) — equal to the two-pass form of layernorm_kernel (norm.hip) to fp32 rounding.
    const float* xr = p.x + row * XC + 4 * hi;
    const float c0 = p.x[row * XC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jb = 0; jb < 16; jb += 8) {                    // two batches of 32 loads (128 VGPRs in flight: nothing else is live yet)
#pragma unroll
        for (int j = jb; j < jb + 8; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4_t v = *(const float4_t*)(xr + 32 * j + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[i] - c0; s1 += d; s2 += d * d; }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    s1 = half_sum(s1); s2 = half_sum(s2);
    if (TR) ts[1] = __builtin_amdgcn_s_memtime();           // statistics pass done (first read of the rows)
    const float m1 = s1 * (1.0f / XC);
    const float mean = c0 + m1;
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / XC) - m1 * m1, 0.f) + p.sub[0].eps);
    __syncthreads();                                        // tables visible
    // ---- second read: Xn fp16 B fragments (k-step ks = 2 j + qp  <-  values 8 qp .. 8 qp + 7 of tile j) and the accumulators' initial
    // value (x + b_out: the residual) ----------------------------------------------------------------------------------------------
    half8_t xn[32];
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
                acc_set<16 * j + 4 * q + i>(v[i] + bo[i]);  // out = (x + b_out) + sum over heads
            });
        });
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);            // batches of 16 loads: the operand registers fill up as the rows turn into them
    });

    // straight-line over the (at most two) sub-layers: a rolled loop carries the accumulators through a phi between the asm walks
    // (accumulator file) and the LayerNorm in between (VALU), which hipcc resolves by spilling 1 679 registers per lane
    xattn_heads<TR>(0, xn, group_sync, nx, voff, hi, p.lk, p.scale_log2, ts, true);
    if (p.nsub > 1) {
        mid_layernorm(xn, lds0 + XTAB + XTABS, p.sub[1].eps, hi);
        xattn_heads<TR>(XNG, xn, group_sync, nx, voff, hi, p.lk, p.scale_log2, ts, false);
    }
    if (TR) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[9] = __builtin_amdgcn_s_memtime(); }                         // all heads
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");        // the last MFMAs may still be in flight and the compiler cannot see them
    wait_vmcnt<0>();                                       // the zero-fill pieces behind the last group (LDS-DMA must not outlive the workgroup)
    // ---- store: row-coalesced through the idle ring ---------------------------------------------------------------------------------
    // A lane owns a token: stored from the accumulators' layout a wave-wide 16-B store touches 32 rows x 32 B — quarter cache lines, the
    // transaction-bound pattern measured on the conv epilogues (here 13.5 k ticks per tile, run 3 of round 6).  The wave
    // dumps half of its tile (32 tokens x 256 channels fp32 = 32 KiB, its quarter of the ring; 16-B piece pc of row r at physical piece
    // pc ^ (r & 7): conflict-free both ways) and reads it back a ROW per instruction: every store is 1 KiB contiguous.
    __syncthreads();                                        // every wave is done reading fragments: the ring is free
    {
        // lane-derived addresses from a FRESH lane id: kept live from the top of the kernel they are what hipcc spills across the head
        // loop (the loop sits at the 256-VGPR limit); the build audit wants no scratch in any shipped kernel
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const int lane = lane_, l32 = lane_ & 31, hi = lane_ >> 5;
        const unsigned wbuf = (unsigned)(size_t)(lptr_t)smem + (unsigned)(wave * XGROUP);
        float* const obase = p.out + (tile0 + wave * 32) * XC + lane * 4;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            static_for<32>([&](auto JQ) {
                constexpr int j = JQ / 4, q = JQ % 4;
                float4_t v;
                if (hh == 0) v = float4_t{acc_get<16 * j + 4 * q>(), acc_get<16 * j + 4 * q + 1>(), acc_get<16 * j + 4 * q + 2>(), acc_get<16 * j + 4 * q + 3>()};
                else v = float4_t{acc_get<128 + 16 * j + 4 * q>(), acc_get<128 + 16 * j + 4 * q + 1>(), acc_get<128 + 16 * j + 4 * q + 2>(), acc_get<128 + 16 * j + 4 * q + 3>()};
                const int pc = 8 * j + 2 * q + hi;
                *(lds_f4wptr_t)(size_t)(wbuf + l32 * 1024 + ((pc ^ (l32 & 7)) << 4)) = v;
            });
            asm volatile("" ::: "memory");                  // (LDS operations of one wave execute in order; the buffer is the wave's own)
#pragma unroll
            for (int kb = 0; kb < 32; kb += 8) {
                float4_t r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = lds_f4(wbuf + (kb + k) * 1024 + ((lane ^ ((kb + k) & 7)) << 4));
#pragma unroll
                for (int k = 0; k < 8; ++k) *(float4_t*)(obase + (long long)(kb + k) * XC + hh * 256) = r[k];
            }
            asm volatile("" ::: "memory");
        }
    }
    if (TR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[10] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long* t = p.trace + (size_t)blockIdx.x * 16;
#pragma unroll
            for (int i = 0; i < 11; ++i) t[i] = ts[i];
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------------------
// Fused FEED-FORWARD sub-layer of BasicTransformerBlock (reference attention.py:562-564 `ff(norm3(x)) + x`; FeedForward / GEGLU of
// diffusers: proj 512 -> 2 x 2048, value * gelu(gate), Linear 2048 -> 512) on the same skeleton:
//
//     out = x + W_down ( (W_v n + b_v) * gelu(W_g n + b_g) ) + b_down ,   n = LayerNorm(x)
//
// The 2 048 hidden channels never exist as a tensor: they are walked in 64 slices of 32, and a slice is shaped like an attention head —
// the "Q" step on W_up rows (value rows of the slice as channel tile 0, gate rows as tile 1: 64 MFMA over k = 512 on the Xn fragments,
// two groups), GEGLU in registers on the D layout (value and gate of one hidden channel sit in the same lane and register), rounded to
// fp16 = the two B fragments of the "to_out" step (W_down[:, slice]: 32 MFMA onto the 256 named accumulators, one group).  Three 32-KiB
// groups per slice, 192 per tile (6 MiB of fragments from L2), one linear stream.  (Slices of 64 — value and gate as two "Q" steps — keep
// 64 fp32 results live beside the 128 operand registers: 4 spilled registers and a 12-minute build.)  Against the two conv-GEMM launches
// it replaces (512 -> 4 096 with the GEGLU epilogue at 0.28 of the MFMA peak: eight k-steps per 256 x 256 tile; 2 048 -> 512): no
// [M][2048] fp16 tensor written and read back, no LayerNorm rows, no tile prologue / epilogue per 512 of k.  Roundings are the chain's:
// LayerNorm rows and the hidden activations fp16, the rest fp32; gelu is uav_gelu_erf of the conv epilogue.  Optionally the result leaves
// as the fp16 hi | lo operand pair of proj_out (cast_f32_hilo_kernel's [M][2 C] rows, bit-identical) instead of / beside the fp32 rows.
constexpr int FTAB_UP = XTAB + XTABS;          // LDS behind gamma | beta | b_down: b_up (value 0 .. 2047 | gate 2048 .. 4095), 16 KiB
constexpr int FSMEM = FTAB_UP + 2 * FINNER * 4;

struct FfArgs {
    const float* x; float* out; half_t* out_hilo;
    const float* gamma; const float* beta; const float* down_bias; const float* up_bias; const char* w; float eps;
};

__global__ __launch_bounds__(256, 1) void ff_sublayer_kernel(FfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const long long tile0 = (long long)blockIdx.x * 128;
    const long long row = tile0 + wave * 32 + l32;

    const unsigned voff = (unsigned)(wave * XPPW * XFRAG + lane * 16);
    const uint4_t wsrd = make_srd(p.w, FNG * XGROUP);
    auto next_of = [&](int s) -> XNext {
        XNext n;
        n.ldsn = lds0 + (unsigned)((s & (XRING - 1)) * XGROUP + wave * XPPW * XFRAG);
        n.srd = wsrd;
        n.so = s < FNG ? (unsigned)s * XGROUP : 0x80000000u;                 // zero-fill pieces behind the last group (see the kernels above)
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
    // ---- tables -> LDS ----------------------------------------------------------------------------------------------------------
    if (tid < 128) {
        *(lds_f4wptr_t)(size_t)(lds0 + XTAB + tid * 16) = ((const float4_t*)p.gamma)[tid];
        *(lds_f4wptr_t)(size_t)(lds0 + XTAB + 2048 + tid * 16) = ((const float4_t*)p.beta)[tid];
        *(lds_f4wptr_t)(size_t)(lds0 + XTAB + 4096 + tid * 16) = ((const float4_t*)p.down_bias)[tid];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *(lds_f4wptr_t)(size_t)(lds0 + FTAB_UP + (k * 256 + tid) * 16) = ((const float4_t*)p.up_bias)[k * 256 + tid];
    // ---- LayerNorm statistics (first read), operand fragments + accumulators (second read): as in the kernels above -------------------
    const float* xr = p.x + row * XC + 4 * hi;
    const float c0 = p.x[row * XC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jb = 0; jb < 16; jb += 8) {
#pragma unroll
        for (int j = jb; j < jb + 8; ++j)
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
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / XC) - m1 * m1, 0.f) + p.eps);
    __syncthreads();                                        // tables visible
    half8_t xn[32];
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
                acc_set<16 * j + 4 * q + i>(v[i] + bo[i]);  // out = (x + b_down) + sum over slices
            });
        });
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    });

    ff_slices(0, xn, group_sync, nx, voff, lane16, lds0 + FTAB_UP);
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    wait_vmcnt<0>();
    __syncthreads();                                        // every wave is done reading fragments: the ring is free
    // ---- store: row-coalesced through the idle ring (see xattn_sublayer_kernel); the hi | lo pair the same way, 8-B pieces ---------------
    {
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const int ln = lane_, l32e = lane_ & 31, hie = lane_ >> 5;
        const unsigned wbuf = (unsigned)(size_t)(lptr_t)smem + (unsigned)(wave * XGROUP);
        if (p.out_hilo) {
            typedef __attribute__((address_space(3))) uint2_t* lds_u2wptr_t;
#pragma unroll
            for (int part = 0; part < 2; ++part) {          // hi = fp16(v), then lo = fp16(v - hi): each 32 rows x 1 KiB in the wave's ring quarter
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
                half_t* const nbase = p.out_hilo + (tile0 + wave * 32) * (2 * XC) + part * XC + ln * 8;
#pragma unroll
                for (int kb = 0; kb < 32; kb += 8) {
                    float4_t r[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) r[k] = lds_f4(wbuf + (kb + k) * 1024 + ((ln ^ ((kb + k) & 7)) << 4));
#pragma unroll
                    for (int k = 0; k < 8; ++k) *(float4_t*)(nbase + (long long)(kb + k) * (2 * XC)) = r[k];
                }
                asm volatile("" ::: "memory");
            }
        }
        if (p.out) {
            float* const obase = p.out + (tile0 + wave * 32) * XC + ln * 4;
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
                    for (int k = 0; k < 8; ++k) *(float4_t*)(obase + (long long)(kb + k) * XC + hh * 256) = r[k];
                }
                asm volatile("" ::: "memory");
            }
        }
    }
}
// Text K | V rows [n_batch * lk][stride] (fp16, head h in columns 64 h ..) -> the fragment stream of the kernel above:
// [n_batch][8 heads][32 fragments][64 lanes][8 halves]; fragments 0 .. 11 = K_h (key tile f % 3, k-step f / 3), 12 .. 23 = V_h^T
// (k-step g >> 1, channel tile g & 1, g = f - 12); keys >= lk and the 8 spare fragments are zero.
__global__ __launch_bounds__(256) void xattn_pack_kv_kernel(const half_t* __restrict__ k, long long k_stride, const half_t* __restrict__ v,
                                                            long long v_stride, int n_batch, int lk, half8_t* __restrict__ out) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;       // one 16-B unit per thread
    const long long total = (long long)n_batch * XHEADS * 32 * 64;
    if (u >= total) return;
    const int lane = (int)(u & 63), f = (int)((u >> 6) & 31), h = (int)((u >> 11) & 7);
    const int bb = (int)(u >> 14);
    const int l32 = lane & 31, hi = lane >> 5;
    half8_t o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (f < 24) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 8 * (e >> 2) + 4 * hi + (e & 3);                 // position inside the 16-wide k-step
            if (f < 12) {
                const int key = 32 * (f % 3) + l32, ch = 16 * (f / 3) + kk;
                if (key < lk) o[e] = k[((long long)bb * lk + key) * k_stride + h * XD + ch];
            } else {
                const int g = f - 12;
                const int key = 16 * (g >> 1) + kk, ch = 32 * (g & 1) + l32;
                if (key < lk) o[e] = v[((long long)bb * lk + key) * v_stride + h * XD + ch];
            }
        }
    }
    out[u] = o;
}

}  // namespace

extern "C" int uav_xattn_pack_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, int32_t n_batch, int32_t lk,
                                 int32_t heads, int32_t head_dim, void* out, void* stream) {
    if (!k || !v || !out) return UAV_EINVAL;
    if (heads != XHEADS || head_dim != XD || n_batch <= 0 || lk <= 0 || lk > 96) return UAV_ESHAPE;
    const long long total = (long long)n_batch * XHEADS * 32 * 64;
    hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)k,
                       (long long)k_stride, (const half_t*)v, (long long)v_stride, n_batch, lk, (half8_t*)out);
    return uav_launch_status();
}

namespace {
int xattn_fill(XattnArgs& a, const float* x, float* out, const uav_xattn_params* subs, int32_t n_subs, int64_t rows, int32_t rows_per_kv,
               int32_t lk, int32_t channels, int32_t heads, float scale) {
    if (!x || !out || !subs) return UAV_EINVAL;
    if (n_subs < 1 || n_subs > 2 || channels != XC || heads != XHEADS || lk <= 0 || lk > 96) return UAV_ESHAPE;
    if (rows <= 0 || rows_per_kv <= 0 || (rows_per_kv % 128) || (rows % rows_per_kv) || rows / 128 >= (1ll << 31)) return UAV_ESHAPE;
    if (((size_t)x | (size_t)out) & 15) return UAV_EALIGN;
    a.x = x; a.out = out; a.nsub = n_subs; a.rows = rows; a.rows_per_kv = rows_per_kv; a.lk = lk;
    a.scale_log2 = scale * 1.44269504088896341f; a.trace = nullptr;
    for (int i = 0; i < 2; ++i) {
        const uav_xattn_params& q = subs[i < n_subs ? i : 0];
        if (!q.ln_gamma || !q.ln_beta || !q.wq_packed || !q.kv_packed || !q.wo_packed || !q.out_bias) return UAV_EINVAL;
        a.sub[i] = XattnSub{q.ln_gamma, q.ln_beta, q.out_bias, (const char*)q.wq_packed, (const char*)q.kv_packed, (const char*)q.wo_packed, q.ln_eps};
    }
    return 0;
}
}  // namespace

extern "C" int uav_xattn_sublayers_f32(const float* x, float* out, const uav_xattn_params* subs, int32_t n_subs, int64_t rows,
                                       int32_t rows_per_kv, int32_t lk, int32_t channels, int32_t heads, float scale, void* stream) {
    XattnArgs a;
    if (int rc = xattn_fill(a, x, out, subs, n_subs, rows, rows_per_kv, lk, channels, heads, scale)) return rc;
    static UavDynLds lds;
    if (int rc = uav_set_dyn_lds(lds, (const void*)xattn_sublayer_kernel<0>, XSMEM)) return rc;
    hipLaunchKernelGGL(xattn_sublayer_kernel<0>, dim3((unsigned)(rows / 128)), dim3(256), XSMEM, (hipStream_t)stream, a);
    return uav_launch_status();
}

extern "C" int uav_ff_sublayer_f32(const float* x, float* out, void* out_hilo, const uav_ff_params* q, int64_t rows, int32_t channels,
                                   int32_t inner, void* stream) {
    if (!x || !q || (!out && !out_hilo) || !q->ln_gamma || !q->ln_beta || !q->w_packed || !q->up_bias || !q->down_bias) return UAV_EINVAL;
    if (channels != XC || inner != FINNER || rows <= 0 || (rows % 128) || rows / 128 >= (1ll << 31)) return UAV_ESHAPE;
    if (((size_t)x | (size_t)out | (size_t)out_hilo | (size_t)q->up_bias | (size_t)q->down_bias | (size_t)q->w_packed) & 15) return UAV_EALIGN;
    FfArgs a{x, out, (half_t*)out_hilo, q->ln_gamma, q->ln_beta, q->down_bias, q->up_bias, (const char*)q->w_packed, q->ln_eps};
    static UavDynLds lds;
    if (int rc = uav_set_dyn_lds(lds, (const void*)ff_sublayer_kernel, FSMEM)) return rc;
    hipLaunchKernelGGL(ff_sublayer_kernel, dim3((unsigned)(rows / 128)), dim3(256), FSMEM, (hipStream_t)stream, a);
    return uav_launch_status();
}

#ifdef UAV_DEV_KERNELS
// Development build only (tools/ab/build_dev.sh): the stamped instance; trace = 16 x uint64 per workgroup (rows / 128 of them).
extern "C" int uav_dev_xattn_sublayers_trace(const float* x, float* out, const uav_xattn_params* subs, int32_t n_subs, int64_t rows,
                                             int32_t rows_per_kv, int32_t lk, float scale, void* trace, void* stream) {
    XattnArgs a;
    if (int rc = xattn_fill(a, x, out, subs, n_subs, rows, rows_per_kv, lk, XC, XHEADS, scale)) return rc;
    a.trace = (unsigned long long*)trace;
    static UavDynLds lds;
    if (int rc = uav_set_dyn_lds(lds, (const void*)xattn_sublayer_kernel<1>, XSMEM)) return rc;
    hipLaunchKernelGGL(xattn_sublayer_kernel<1>, dim3((unsigned)(rows / 128)), dim3(256), XSMEM, (hipStream_t)stream, a);
    return uav_launch_status();
}
#endif
