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
#include "uav_common.h"

namespace {

constexpr int XC = 512, XHEADS = 8, XD = 64;
constexpr int XFRAG = 1024;                    // bytes of one A fragment (32 rows x 16 k, fp16)
constexpr int XGROUP = 32 * XFRAG;             // one ring slot: 32 fragments
constexpr int XRING = 4;                       // groups resident in LDS
constexpr int XGPH = 5;                        // groups per head: W_q (2), K | V^T (1), W_out (2)
constexpr int XNG = XHEADS * XGPH;             // groups per tile
constexpr int XPPW = 8;                        // 1-KiB DMA pieces per wave and group
constexpr int XTAB = XRING * XGROUP;           // LDS offset of gamma | beta | bias (3 x 2 KiB)
constexpr int XSMEM = XTAB + 3 * XC * 4;

struct XattnArgs {
    const float* x; float* out; const float* gamma; const float* beta; const float* bias;
    const char* wq; const char* kv; const char* wo;
    long long rows; int rows_per_kv; int lk; float eps, scale_log2;
    unsigned long long* trace;                 // development instance only (UAV_DEV_KERNELS): 16 s_memtime stamps per workgroup
};

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) const float4_t* lds_f4ptr_t;
typedef __attribute__((address_space(3))) float4_t* lds_f4wptr_t;

UAV_DEVINL float4_t lds_f4(unsigned a) { return *(lds_f4ptr_t)(size_t)a; }

UAV_DEVINL uint4_t make_srd(const char* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    uint4_t r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
// one 1-KiB piece: lane l fetches 16 B at srd.base + voff(l) + soff and the hardware drops it at LDS m0 + 16 l
UAV_DEVINL void dma_piece(uint4_t srd, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %[l]\n"
                 "s_nop 0\n"
                 "buffer_load_dwordx4 %[v], %[s], %[o] offen lds\n"
                 :: [l] "s"(lds_dst), [v] "v"(voff), [s] "s"(srd), [o] "s"(soff) : "memory");
}
template <int N> UAV_DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
UAV_DEVINL float swap32(float v) { return __shfl_xor(v, 32, 64); }

// ---- the asm walk of a group: XRD = read fragment into t, XS = wait for the oldest read, MFMA on it, refill its register,
// XT = the same without a refill (tail).  Fragment f of a group sits at byte f * 1024 (+ 16 * lane) of the slot.
#define XRD(T, OFF) "ds_read_b128 %[" #T "], %[st] offset:" #OFF "\n"
#define XMF(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], %[" #C "]\n"
#define XS(T, C, B, WN, OFF) "s_waitcnt lgkmcnt(" #WN ")\n" XMF(C, T, B) XRD(T, OFF)
#define XT(T, C, B, WN) "s_waitcnt lgkmcnt(" #WN ")\n" XMF(C, T, B)
// the first MFMA on an accumulator: C = the inline constant 0 (the accumulator is a pure output: nothing to zero, no zero tuple kept live)
#define XMF0(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], 0\n"
#define XS0(T, C, B, WN, OFF) "s_waitcnt lgkmcnt(" #WN ")\n" XMF0(C, T, B) XRD(T, OFF)
#define XT0(T, C, B, WN) "s_waitcnt lgkmcnt(" #WN ")\n" XMF0(C, T, B)
// one 1-KiB LDS-DMA piece of the group three ahead, between two MFMAs: 16 B per lane from srd.base + voff + so + GOFF to LDS
// M0 + GOFF + 16 lane — the 12-bit instruction offset moves BOTH addresses (the first interleaved version set M0 to the piece's own
// place and added GOFF on top: pieces 1-3 of every half landed 1-3 KiB too far, NaN; run 2 of round 6) —, so M0 = the half group's
// base; XDADV steps `so` over the four pieces addressed through the immediate
#define XD(LOFF, GOFF) "s_add_u32 m0, %[ldsn], " #LOFF "\n" "s_nop 0\n" "buffer_load_dwordx4 %[voff], %[srd], %[so] offen offset:" #GOFF " lds\n"
#define XDADV "s_add_u32 %[so], %[so], 4096\n"
// W_q group: fragment f = (k-step f >> 1, channel tile f & 1);  W_out group: (channel tile 2 (f >> 3) + (f & 1), k-step (f >> 1) & 3);
// K: (key tile f % 3, k-step f / 3);  V^T (fragments 12 .. 23 of the K | V group): (k-step f >> 1, channel tile f & 1) — consecutive
// MFMAs never share an accumulator.
#define XG_WQ_FIRST \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XS0(t0, q0, b0, 5, 6144) \
    XS0(t1, q1, b0, 5, 7168) XS(t2, q0, b1, 5, 8192) XD(0, 0) XS(t3, q1, b1, 5, 9216) XS(t4, q0, b2, 5, 10240) \
    XS(t5, q1, b2, 5, 11264) XS(t0, q0, b3, 5, 12288) XD(0, 1024) XS(t1, q1, b3, 5, 13312) XS(t2, q0, b4, 5, 14336) \
    XS(t3, q1, b4, 5, 15360) XS(t4, q0, b5, 5, 16384) XD(0, 2048) XS(t5, q1, b5, 5, 17408) XS(t0, q0, b6, 5, 18432) \
    XS(t1, q1, b6, 5, 19456) XS(t2, q0, b7, 5, 20480) XD(0, 3072) XDADV XS(t3, q1, b7, 5, 21504) XS(t4, q0, b8, 5, 22528) \
    XS(t5, q1, b8, 5, 23552) XS(t0, q0, b9, 5, 24576) XD(4096, 0) XS(t1, q1, b9, 5, 25600) XS(t2, q0, b10, 5, 26624) \
    XS(t3, q1, b10, 5, 27648) XS(t4, q0, b11, 5, 28672) XD(4096, 1024) XS(t5, q1, b11, 5, 29696) \
    XS(t0, q0, b12, 5, 30720) XS(t1, q1, b12, 5, 31744) XT(t2, q0, b13, 5) XD(4096, 2048) XT(t3, q1, b13, 4) \
    XT(t4, q0, b14, 3) XT(t5, q1, b14, 2) XT(t0, q0, b15, 1) XD(4096, 3072) XT(t1, q1, b15, 0)

#define XG_WQ \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XS(t0, q0, b0, 5, 6144) \
    XS(t1, q1, b0, 5, 7168) XS(t2, q0, b1, 5, 8192) XD(0, 0) XS(t3, q1, b1, 5, 9216) XS(t4, q0, b2, 5, 10240) \
    XS(t5, q1, b2, 5, 11264) XS(t0, q0, b3, 5, 12288) XD(0, 1024) XS(t1, q1, b3, 5, 13312) XS(t2, q0, b4, 5, 14336) \
    XS(t3, q1, b4, 5, 15360) XS(t4, q0, b5, 5, 16384) XD(0, 2048) XS(t5, q1, b5, 5, 17408) XS(t0, q0, b6, 5, 18432) \
    XS(t1, q1, b6, 5, 19456) XS(t2, q0, b7, 5, 20480) XD(0, 3072) XDADV XS(t3, q1, b7, 5, 21504) XS(t4, q0, b8, 5, 22528) \
    XS(t5, q1, b8, 5, 23552) XS(t0, q0, b9, 5, 24576) XD(4096, 0) XS(t1, q1, b9, 5, 25600) XS(t2, q0, b10, 5, 26624) \
    XS(t3, q1, b10, 5, 27648) XS(t4, q0, b11, 5, 28672) XD(4096, 1024) XS(t5, q1, b11, 5, 29696) \
    XS(t0, q0, b12, 5, 30720) XS(t1, q1, b12, 5, 31744) XT(t2, q0, b13, 5) XD(4096, 2048) XT(t3, q1, b13, 4) \
    XT(t4, q0, b14, 3) XT(t5, q1, b14, 2) XT(t0, q0, b15, 1) XD(4096, 3072) XT(t1, q1, b15, 0)

#define XG_WO \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XS(t0, c0, b0, 5, 6144) \
    XS(t1, c1, b0, 5, 7168) XS(t2, c0, b1, 5, 8192) XD(0, 0) XS(t3, c1, b1, 5, 9216) XS(t4, c0, b2, 5, 10240) \
    XS(t5, c1, b2, 5, 11264) XS(t0, c0, b3, 5, 12288) XD(0, 1024) XS(t1, c1, b3, 5, 13312) XS(t2, c2, b0, 5, 14336) \
    XS(t3, c3, b0, 5, 15360) XS(t4, c2, b1, 5, 16384) XD(0, 2048) XS(t5, c3, b1, 5, 17408) XS(t0, c2, b2, 5, 18432) \
    XS(t1, c3, b2, 5, 19456) XS(t2, c2, b3, 5, 20480) XD(0, 3072) XDADV XS(t3, c3, b3, 5, 21504) XS(t4, c4, b0, 5, 22528) \
    XS(t5, c5, b0, 5, 23552) XS(t0, c4, b1, 5, 24576) XD(4096, 0) XS(t1, c5, b1, 5, 25600) XS(t2, c4, b2, 5, 26624) \
    XS(t3, c5, b2, 5, 27648) XS(t4, c4, b3, 5, 28672) XD(4096, 1024) XS(t5, c5, b3, 5, 29696) XS(t0, c6, b0, 5, 30720) \
    XS(t1, c7, b0, 5, 31744) XT(t2, c6, b1, 5) XD(4096, 2048) XT(t3, c7, b1, 4) XT(t4, c6, b2, 3) XT(t5, c7, b2, 2) \
    XT(t0, c6, b3, 1) XD(4096, 3072) XT(t1, c7, b3, 0)

#define XG_K \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XS0(t0, c0, b0, 5, 6144) \
    XS0(t1, c1, b0, 5, 7168) XD(0, 0) XS0(t2, c2, b0, 5, 8192) XS(t3, c0, b1, 5, 9216) XS(t4, c1, b1, 5, 10240) \
    XD(0, 1024) XS(t5, c2, b1, 5, 11264) XT(t0, c0, b2, 5) XT(t1, c1, b2, 4) XD(0, 2048) XT(t2, c2, b2, 3) \
    XT(t3, c0, b3, 2) XT(t4, c1, b3, 1) XD(0, 3072) XDADV XT(t5, c2, b3, 0)

#define XG_V \
    XRD(t0, 12288) XRD(t1, 13312) XRD(t2, 14336) XRD(t3, 15360) XRD(t4, 16384) XRD(t5, 17408) XS0(t0, c0, b0, 5, 18432) \
    XS0(t1, c1, b0, 5, 19456) XD(4096, 0) XS(t2, c0, b1, 5, 20480) XS(t3, c1, b1, 5, 21504) XS(t4, c0, b2, 5, 22528) \
    XD(4096, 1024) XS(t5, c1, b2, 5, 23552) XT(t0, c0, b3, 5) XT(t1, c1, b3, 4) XD(4096, 2048) XT(t2, c0, b4, 3) \
    XT(t3, c1, b4, 2) XT(t4, c0, b5, 1) XD(4096, 3072) XT(t5, c1, b5, 0)
#define XTMP_OUT [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [so] "+s"(nx.so)
#define XDMA_IN [ldsn] "s"(nx.ldsn), [srd] "s"(nx.srd), [voff] "v"(voff)

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

    // ---- the stream of A fragments: 40 groups of 32 KiB, eight 1-KiB pieces per wave and group ---------------------------------
    const uint4_t srd_wq = make_srd(p.wq, XHEADS * 2 * XGROUP);
    const uint4_t srd_wo = make_srd(p.wo, XHEADS * 2 * XGROUP);
    const uint4_t srd_kv = make_srd(p.kv + (long long)b * XHEADS * XGROUP, XHEADS * XGROUP);
    const unsigned voff = (unsigned)(wave * XPPW * XFRAG + lane * 16);
    struct Next { uint4_t srd; unsigned so, ldsn; };       // the group XRING - 1 = 3 ahead: its source and its ring slot
    auto next_of = [&](int h, int j) -> Next {            // (h, j): group j (0, 1: W_q; 2: K | V; 3, 4: W_out) of head h
        Next n;
        const int s = h * XGPH + j;
        n.ldsn = lds0 + (unsigned)((s & (XRING - 1)) * XGROUP + wave * XPPW * XFRAG);
        if (j < 2) { n.srd = srd_wq; n.so = (unsigned)((h * 2 + j) * XGROUP); }
        else if (j == 2) { n.srd = srd_kv; n.so = (unsigned)(h * XGROUP); }
        else { n.srd = srd_wo; n.so = (unsigned)((h * 2 + (j - 3)) * XGROUP); }
        // behind the last group of the tile: the same eight pieces with every lane out of the descriptor's range — the hardware fetches
        // nothing and zero-fills a slot nobody reads again, and the vmcnt arithmetic below stays the same for every group
        if (s >= XNG) n.so = 0x80000000u;
        return n;
    };
    auto issue = [&](int h, int j) {                       // prologue: a whole group at once
        Next n = next_of(h, j);
#pragma unroll
        for (int i = 0; i < XPPW; ++i) dma_piece(n.srd, voff, n.so + i * XFRAG, n.ldsn + i * XFRAG);
    };
    // before group (h, j) is read: this wave's pieces of it have landed (the two groups issued behind it may still fly: 16 pieces),
    // then every wave's have (barrier) — which also says every wave is done with the group before it, whose slot the group three
    // ahead is written into WHILE this group is multiplied (the pieces sit between the MFMAs of the asm walk).
    Next nx;
    auto group_sync = [&](int h, int j) -> unsigned {
        wait_vmcnt<XPPW * (XRING - 2)>();
        __syncthreads();
        const int jn = j + XRING - 1;                      // (h, j + 3) or (h + 1, j - 2)
        nx = next_of(jn < XGPH ? h : h + 1, jn < XGPH ? jn : jn - XGPH);
        return lds0 + (unsigned)(((h * XGPH + j) & (XRING - 1)) * XGROUP) + lane * 16;
    };

#pragma unroll
    for (int s = 0; s < XRING - 1; ++s) issue(0, s);
    // ---- tables -> LDS; LayerNorm statistics of the lane's token ------------------------------------------------------------------
    if (tid < 128) {
        *(lds_f4wptr_t)(size_t)(lds0 + XTAB + tid * 16) = ((const float4_t*)p.gamma)[tid];
        *(lds_f4wptr_t)(size_t)(lds0 + XTAB + 2048 + tid * 16) = ((const float4_t*)p.beta)[tid];
        *(lds_f4wptr_t)(size_t)(lds0 + XTAB + 4096 + tid * 16) = ((const float4_t*)p.bias)[tid];
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
    s1 += swap32(s1); s2 += swap32(s2);
    if (TR) ts[1] = __builtin_amdgcn_s_memtime();           // statistics pass done (first read of the rows)
    const float m1 = s1 * (1.0f / XC);
    const float mean = c0 + m1;
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / XC) - m1 * m1, 0.f) + p.eps);
    __syncthreads();                                        // tables visible
    // ---- second read: Xn fp16 B fragments (k-step ks = 2 j + qp  <-  values 8 qp .. 8 qp + 7 of tile j) and the accumulators' initial
    // value (x + b_out: the residual) ----------------------------------------------------------------------------------------------
    float16_t acc[16];
    half8_t xn[32];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4_t v = *(const float4_t*)(xr + 32 * j + 8 * q);
            const unsigned ta = lds0 + XTAB + (32 * j + 8 * q + 4 * hi) * 4;
            const float4_t g = lds_f4(ta), be = lds_f4(ta + 2048), bo = lds_f4(ta + 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xn[2 * j + (q >> 1)][4 * (q & 1) + i] = (half_t)((v[i] - mean) * rstd * g[i] + be[i]);
                acc[j][4 * q + i] = v[i] + bo[i];           // out = (x + b_out) + sum over heads
            }
        }
        if (j == 7 || j == 11) __builtin_amdgcn_sched_barrier(0);       // batches of 32, 16, 16 loads: the operand registers fill up as the rows turn into them
    }

    if (TR) ts[2] = __builtin_amdgcn_s_memtime();           // operand fragments and accumulators built (second read)
    // ---- heads ----------------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int h = 0; h < XHEADS; ++h) {
        half8_t t0, t1, t2, t3, t4, t5;
        if (TR && h == 1) ts[3] = __builtin_amdgcn_s_memtime();    // head 1 is stamped phase by phase (head 0 carries the cold start)
        // Q_h^T [64 ch][32 tokens] = Wq_h . Xn^T
        float16_t q0, q1;
        {
            const unsigned st = group_sync(h, 0);
            const int j = 0;
            asm volatile(XG_WQ_FIRST : [q0] "=&v"(q0), [q1] "=&v"(q1), XTMP_OUT
                         : [st] "v"(st), [b0] "v"(xn[16 * j + 0]), [b1] "v"(xn[16 * j + 1]), [b2] "v"(xn[16 * j + 2]), [b3] "v"(xn[16 * j + 3]),
                           [b4] "v"(xn[16 * j + 4]), [b5] "v"(xn[16 * j + 5]), [b6] "v"(xn[16 * j + 6]), [b7] "v"(xn[16 * j + 7]),
                           [b8] "v"(xn[16 * j + 8]), [b9] "v"(xn[16 * j + 9]), [b10] "v"(xn[16 * j + 10]), [b11] "v"(xn[16 * j + 11]),
                           [b12] "v"(xn[16 * j + 12]), [b13] "v"(xn[16 * j + 13]), [b14] "v"(xn[16 * j + 14]), [b15] "v"(xn[16 * j + 15]), XDMA_IN
                         : "memory", "scc");
        }
        {
            const int j = 1;
            const unsigned st = group_sync(h, j);
            asm volatile(XG_WQ : [q0] "+v"(q0), [q1] "+v"(q1), XTMP_OUT
                         : [st] "v"(st), [b0] "v"(xn[16 * j + 0]), [b1] "v"(xn[16 * j + 1]), [b2] "v"(xn[16 * j + 2]), [b3] "v"(xn[16 * j + 3]),
                           [b4] "v"(xn[16 * j + 4]), [b5] "v"(xn[16 * j + 5]), [b6] "v"(xn[16 * j + 6]), [b7] "v"(xn[16 * j + 7]),
                           [b8] "v"(xn[16 * j + 8]), [b9] "v"(xn[16 * j + 9]), [b10] "v"(xn[16 * j + 10]), [b11] "v"(xn[16 * j + 11]),
                           [b12] "v"(xn[16 * j + 12]), [b13] "v"(xn[16 * j + 13]), [b14] "v"(xn[16 * j + 14]), [b15] "v"(xn[16 * j + 15]), XDMA_IN
                         : "memory", "scc");
        }
        if (TR && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[4] = __builtin_amdgcn_s_memtime(); }     // Q GEMM (64 MFMA)
        half8_t qf[4];                                      // Q rounded to fp16 like the stored q of the unfused chain
#pragma unroll
        for (int e = 0; e < 8; ++e) { qf[0][e] = (half_t)q0[e]; qf[1][e] = (half_t)q0[8 + e]; qf[2][e] = (half_t)q1[e]; qf[3][e] = (half_t)q1[8 + e]; }
        // S^T [96 keys][32 tokens] = K_h . Q^T
        float16_t sacc[3];
        const unsigned stkv = group_sync(h, 2);
        asm volatile(XG_K : [c0] "=&v"(sacc[0]), [c1] "=&v"(sacc[1]), [c2] "=&v"(sacc[2]), XTMP_OUT
                     : [st] "v"(stkv), [b0] "v"(qf[0]), [b1] "v"(qf[1]), [b2] "v"(qf[2]), [b3] "v"(qf[3]), XDMA_IN : "memory", "scc");
        if (TR && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[5] = __builtin_amdgcn_s_memtime(); }     // S = K Q (12 MFMA)
        // softmax over the keys: this lane holds keys 32 t + (r & 3) + 8 (r >> 2) + 4 hi, lane ^ 32 the others
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float s = sacc[t][r] * p.scale_log2;
                s = key < p.lk ? s : -INFINITY;
                sacc[t][r] = s; mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, swap32(mx));
        float ps = 0.f;
        half8_t pf[6];                                      // P^T B fragments: k-step 2 t + (r >> 3)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sacc[t][r] - mx);
                ps += e;
                pf[2 * t + (r >> 3)][r & 7] = (half_t)e;
            }
            __builtin_amdgcn_sched_barrier(0);              // one key tile at a time: hipcc otherwise keeps all 48 exponentials in fp32 beside S and P
        }
        ps += swap32(ps);
        const float inv = 1.0f / ps;
        if (TR && h == 1) ts[6] = __builtin_amdgcn_s_memtime();                                                          // softmax
        // O^T [64 ch][32 tokens] = V_h^T . P^T (same LDS slot, fragments 12 .. 23)
        float16_t o0, o1;
        asm volatile(XG_V : [c0] "=&v"(o0), [c1] "=&v"(o1), XTMP_OUT
                     : [st] "v"(stkv), [b0] "v"(pf[0]), [b1] "v"(pf[1]), [b2] "v"(pf[2]), [b3] "v"(pf[3]), [b4] "v"(pf[4]), [b5] "v"(pf[5]), XDMA_IN
                     : "memory", "scc");
        if (TR && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[7] = __builtin_amdgcn_s_memtime(); }     // O = V P (12 MFMA)
        half8_t of[4];                                      // O / l rounded to fp16 like the stored attention output
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            of[0][e] = (half_t)(o0[e] * inv); of[1][e] = (half_t)(o0[8 + e] * inv);
            of[2][e] = (half_t)(o1[e] * inv); of[3][e] = (half_t)(o1[8 + e] * inv);
        }
        // acc [512 ch][32 tokens] += Wout[:, head h] . O^T
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned st = group_sync(h, 3 + j);
            asm volatile(XG_WO : [c0] "+a"(acc[8 * j + 0]), [c1] "+a"(acc[8 * j + 1]), [c2] "+a"(acc[8 * j + 2]), [c3] "+a"(acc[8 * j + 3]),
                           [c4] "+a"(acc[8 * j + 4]), [c5] "+a"(acc[8 * j + 5]), [c6] "+a"(acc[8 * j + 6]), [c7] "+a"(acc[8 * j + 7]), XTMP_OUT
                         : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), [b2] "v"(of[2]), [b3] "v"(of[3]), XDMA_IN : "memory", "scc");
        }
        if (TR && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[8] = __builtin_amdgcn_s_memtime(); }     // acc += Wout O (64 MFMA)
    }
    if (TR) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[9] = __builtin_amdgcn_s_memtime(); }                         // all heads
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
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4_t v = {acc[8 * hh + j][4 * q], acc[8 * hh + j][4 * q + 1], acc[8 * hh + j][4 * q + 2], acc[8 * hh + j][4 * q + 3]};
                    const int pc = 8 * j + 2 * q + hi;
                    *(lds_f4wptr_t)(size_t)(wbuf + l32 * 1024 + ((pc ^ (l32 & 7)) << 4)) = v;
                }
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

extern "C" int uav_xattn_sublayer_f32(const float* x, float* out, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                      const void* wq_packed, const void* kv_packed, const void* wo_packed, const float* out_bias,
                                      int64_t rows, int32_t rows_per_kv, int32_t lk, int32_t channels, int32_t heads, float scale,
                                      void* stream) {
    if (!x || !out || !ln_gamma || !ln_beta || !wq_packed || !kv_packed || !wo_packed || !out_bias) return UAV_EINVAL;
    if (channels != XC || heads != XHEADS || lk <= 0 || lk > 96) return UAV_ESHAPE;
    if (rows <= 0 || rows_per_kv <= 0 || (rows_per_kv % 128) || (rows % rows_per_kv) || rows / 128 >= (1ll << 31)) return UAV_ESHAPE;
    if (((size_t)x | (size_t)out) & 15) return UAV_EALIGN;
    static UavDynLds lds;
    if (int rc = uav_set_dyn_lds(lds, (const void*)xattn_sublayer_kernel<0>, XSMEM)) return rc;
    XattnArgs a{x, out, ln_gamma, ln_beta, out_bias, (const char*)wq_packed, (const char*)kv_packed, (const char*)wo_packed,
                (long long)rows, rows_per_kv, lk, ln_eps, scale * 1.44269504088896341f, nullptr};
    hipLaunchKernelGGL(xattn_sublayer_kernel<0>, dim3((unsigned)(rows / 128)), dim3(256), XSMEM, (hipStream_t)stream, a);
    return uav_launch_status();
}

#ifdef UAV_DEV_KERNELS
// Development build only (tools/ab/build_dev.sh): the stamped instance; trace = 16 x uint64 per workgroup (rows / 128 of them).
extern "C" int uav_dev_xattn_sublayer_trace(const float* x, float* out, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                            const void* wq_packed, const void* kv_packed, const void* wo_packed, const float* out_bias,
                                            int64_t rows, int32_t rows_per_kv, int32_t lk, float scale, void* trace, void* stream) {
    static UavDynLds lds;
    if (int rc = uav_set_dyn_lds(lds, (const void*)xattn_sublayer_kernel<1>, XSMEM)) return rc;
    XattnArgs a{x, out, ln_gamma, ln_beta, out_bias, (const char*)wq_packed, (const char*)kv_packed, (const char*)wo_packed,
                (long long)rows, rows_per_kv, lk, ln_eps, scale * 1.44269504088896341f, (unsigned long long*)trace};
    hipLaunchKernelGGL(xattn_sublayer_kernel<1>, dim3((unsigned)(rows / 128)), dim3(256), XSMEM, (hipStream_t)stream, a);
    return uav_launch_status();
}
#endif
