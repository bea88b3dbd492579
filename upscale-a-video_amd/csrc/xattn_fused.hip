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
#include <utility>

namespace {

constexpr int XC = 512, XHEADS = 8, XD = 64;
constexpr int XFRAG = 1024;                    // bytes of one A fragment (32 rows x 16 k, fp16)
constexpr int XGROUP = 32 * XFRAG;             // one ring slot: 32 fragments
constexpr int XRING = 4;                       // groups resident in LDS
constexpr int XGPH = 5;                        // groups per head: W_q (2), K | V^T (1), W_out (2)
constexpr int XNG = XHEADS * XGPH;             // groups per tile
constexpr int XPPW = 8;                        // 1-KiB DMA pieces per wave and group
constexpr int XTAB = XRING * XGROUP;           // LDS offset of gamma | beta | bias (3 x 2 KiB) of the first sub-layer, then of the second
constexpr int XTABS = 3 * XC * 4;
constexpr int XSMEM = XTAB + 2 * XTABS;

struct XattnSub {                              // one sub-layer: its LayerNorm, its packed projections, the text K | V of its to_k / to_v
    const float* gamma; const float* beta; const float* bias;
    const char* wq; const char* kv; const char* wo; float eps;
};
struct XattnArgs {
    const float* x; float* out;
    XattnSub sub[2]; int nsub;                 // 1, or 2 consecutive sub-layers of one block (attn1 with only_cross_attention, then attn2)
    long long rows; int rows_per_kv; int lk; float scale_log2;
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
UAV_DEVINL uint32_t pack_h2f(float x, float y) {
    half2_t h = {(half_t)x, (half_t)y};
    return __builtin_bit_cast(uint32_t, h);
}

// The 256 fp32 accumulators of a wave's 32 tokens x 512 channels live in the accumulator half of the register file BY NAME — channel
// tile nt in a[16 nt : 16 nt + 15] — like the O^T tile of attn512w_kernel (attention.hip): as C++ tuples that asm statements take as
// "+a" operands AND the VALU touches (residual in, second LayerNorm, store) hipcc shuffled them between the two halves and spilled 34 ...
// 1 679 registers per lane.  Every statement that names them lists the whole accumulator file as clobbered — that also makes the kernel
// descriptor allocate it — and the compiler never uses AGPRs itself (build audit: uav/build.py audit_accumulator_file).
#define XACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
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
// accumulator N <- v / -> v (N a compile-time constant: the callers unroll over std::integral_constant)
template <int N> UAV_DEVINL void acc_set(float v) { asm volatile("v_accvgpr_write_b32 a%c0, %1" :: "i"(N), "v"(v) : XACC_CLOBBERS); }
template <int N> UAV_DEVINL float acc_get() { float v; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(N) : XACC_CLOBBERS); return v; }
template <int... I, class F> UAV_DEVINL void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> UAV_DEVINL void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- the asm walk of a group: XRD = read fragment into t, XS = wait for the oldest read, MFMA on it, refill its register,
// XT = the same without a refill (tail).  Fragment f of a group sits at byte f * 1024 (+ 16 * lane) of the slot.
#define XRD(T, OFF) "ds_read_b128 %[" #T "], %[st] offset:" #OFF "\n"
#define XMF(C, A, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #A "], %[" #B "], %[" #C "]\n"
#define XS(T, C, B, WN, OFF) "s_waitcnt lgkmcnt(" #WN ")\n" XMF(C, T, B) XRD(T, OFF)
#define XT(T, C, B, WN) "s_waitcnt lgkmcnt(" #WN ")\n" XMF(C, T, B)
// the operands the other way round (A = the register fragment B, B = the LDS fragment T): D[token][channel]
#define XMFU(C, T, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #B "], %[" #T "], %[" #C "]\n"
#define XMFU0(C, T, B) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #B "], %[" #T "], 0\n"
#define XSU(T, C, B, WN, OFF) "s_waitcnt lgkmcnt(" #WN ")\n" XMFU(C, T, B) XRD(T, OFF)
#define XTU(T, C, B, WN) "s_waitcnt lgkmcnt(" #WN ")\n" XMFU(C, T, B)
#define XSU0(T, C, B, WN, OFF) "s_waitcnt lgkmcnt(" #WN ")\n" XMFU0(C, T, B) XRD(T, OFF)
#define XTU0(T, C, B, WN) "s_waitcnt lgkmcnt(" #WN ")\n" XMFU0(C, T, B)
// behind the last MFMA of a group whose accumulators the VALU reads next: the compiler cannot see MFMAs inside an asm statement and
// inserts none of the wait states their results need
#define XNOP "s_nop 15\ns_nop 3\n"
// MFMA on a NAMED accumulator tile a[LO:HI] (the 256 output accumulators, see XACC_CLOBBERS)
#define XMFA(LO, HI, A, B) "v_mfma_f32_32x32x16_f16 a[" #LO ":" #HI "], %[" #A "], %[" #B "], a[" #LO ":" #HI "]\n"
#define XSA(T, LO, HI, B, WN, OFF) "s_waitcnt lgkmcnt(" #WN ")\n" XMFA(LO, HI, T, B) XRD(T, OFF)
#define XTA(T, LO, HI, B, WN) "s_waitcnt lgkmcnt(" #WN ")\n" XMFA(LO, HI, T, B)
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
    XT(t4, q0, b14, 3) XT(t5, q1, b14, 2) XT(t0, q0, b15, 1) XD(4096, 3072) XT(t1, q1, b15, 0) XNOP

#define XG_WV_FIRST \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XSU0(t0, q0, b0, 5, 6144) \
    XSU0(t1, q1, b0, 5, 7168) XSU(t2, q0, b1, 5, 8192) XD(0, 0) XSU(t3, q1, b1, 5, 9216) XSU(t4, q0, b2, 5, 10240) \
    XSU(t5, q1, b2, 5, 11264) XSU(t0, q0, b3, 5, 12288) XD(0, 1024) XSU(t1, q1, b3, 5, 13312) XSU(t2, q0, b4, 5, 14336) \
    XSU(t3, q1, b4, 5, 15360) XSU(t4, q0, b5, 5, 16384) XD(0, 2048) XSU(t5, q1, b5, 5, 17408) XSU(t0, q0, b6, 5, 18432) \
    XSU(t1, q1, b6, 5, 19456) XSU(t2, q0, b7, 5, 20480) XD(0, 3072) XDADV XSU(t3, q1, b7, 5, 21504) \
    XSU(t4, q0, b8, 5, 22528) XSU(t5, q1, b8, 5, 23552) XSU(t0, q0, b9, 5, 24576) XD(4096, 0) XSU(t1, q1, b9, 5, 25600) \
    XSU(t2, q0, b10, 5, 26624) XSU(t3, q1, b10, 5, 27648) XSU(t4, q0, b11, 5, 28672) XD(4096, 1024) \
    XSU(t5, q1, b11, 5, 29696) XSU(t0, q0, b12, 5, 30720) XSU(t1, q1, b12, 5, 31744) XTU(t2, q0, b13, 5) XD(4096, 2048) \
    XTU(t3, q1, b13, 4) XTU(t4, q0, b14, 3) XTU(t5, q1, b14, 2) XTU(t0, q0, b15, 1) XD(4096, 3072) XTU(t1, q1, b15, 0)

#define XG_WV \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XSU(t0, q0, b0, 5, 6144) \
    XSU(t1, q1, b0, 5, 7168) XSU(t2, q0, b1, 5, 8192) XD(0, 0) XSU(t3, q1, b1, 5, 9216) XSU(t4, q0, b2, 5, 10240) \
    XSU(t5, q1, b2, 5, 11264) XSU(t0, q0, b3, 5, 12288) XD(0, 1024) XSU(t1, q1, b3, 5, 13312) XSU(t2, q0, b4, 5, 14336) \
    XSU(t3, q1, b4, 5, 15360) XSU(t4, q0, b5, 5, 16384) XD(0, 2048) XSU(t5, q1, b5, 5, 17408) XSU(t0, q0, b6, 5, 18432) \
    XSU(t1, q1, b6, 5, 19456) XSU(t2, q0, b7, 5, 20480) XD(0, 3072) XDADV XSU(t3, q1, b7, 5, 21504) \
    XSU(t4, q0, b8, 5, 22528) XSU(t5, q1, b8, 5, 23552) XSU(t0, q0, b9, 5, 24576) XD(4096, 0) XSU(t1, q1, b9, 5, 25600) \
    XSU(t2, q0, b10, 5, 26624) XSU(t3, q1, b10, 5, 27648) XSU(t4, q0, b11, 5, 28672) XD(4096, 1024) \
    XSU(t5, q1, b11, 5, 29696) XSU(t0, q0, b12, 5, 30720) XSU(t1, q1, b12, 5, 31744) XTU(t2, q0, b13, 5) XD(4096, 2048) \
    XTU(t3, q1, b13, 4) XTU(t4, q0, b14, 3) XTU(t5, q1, b14, 2) XTU(t0, q0, b15, 1) XD(4096, 3072) XTU(t1, q1, b15, 0) \
    XNOP

#define XG_WO0 \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XSA(t0, 0, 15, b0, 5, 6144) \
    XSA(t1, 16, 31, b0, 5, 7168) XSA(t2, 0, 15, b1, 5, 8192) XD(0, 0) XSA(t3, 16, 31, b1, 5, 9216) \
    XSA(t4, 0, 15, b2, 5, 10240) XSA(t5, 16, 31, b2, 5, 11264) XSA(t0, 0, 15, b3, 5, 12288) XD(0, 1024) \
    XSA(t1, 16, 31, b3, 5, 13312) XSA(t2, 32, 47, b0, 5, 14336) XSA(t3, 48, 63, b0, 5, 15360) \
    XSA(t4, 32, 47, b1, 5, 16384) XD(0, 2048) XSA(t5, 48, 63, b1, 5, 17408) XSA(t0, 32, 47, b2, 5, 18432) \
    XSA(t1, 48, 63, b2, 5, 19456) XSA(t2, 32, 47, b3, 5, 20480) XD(0, 3072) XDADV XSA(t3, 48, 63, b3, 5, 21504) \
    XSA(t4, 64, 79, b0, 5, 22528) XSA(t5, 80, 95, b0, 5, 23552) XSA(t0, 64, 79, b1, 5, 24576) XD(4096, 0) \
    XSA(t1, 80, 95, b1, 5, 25600) XSA(t2, 64, 79, b2, 5, 26624) XSA(t3, 80, 95, b2, 5, 27648) \
    XSA(t4, 64, 79, b3, 5, 28672) XD(4096, 1024) XSA(t5, 80, 95, b3, 5, 29696) XSA(t0, 96, 111, b0, 5, 30720) \
    XSA(t1, 112, 127, b0, 5, 31744) XTA(t2, 96, 111, b1, 5) XD(4096, 2048) XTA(t3, 112, 127, b1, 4) \
    XTA(t4, 96, 111, b2, 3) XTA(t5, 112, 127, b2, 2) XTA(t0, 96, 111, b3, 1) XD(4096, 3072) XTA(t1, 112, 127, b3, 0)

#define XG_WO1 \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XSA(t0, 128, 143, b0, 5, 6144) \
    XSA(t1, 144, 159, b0, 5, 7168) XSA(t2, 128, 143, b1, 5, 8192) XD(0, 0) XSA(t3, 144, 159, b1, 5, 9216) \
    XSA(t4, 128, 143, b2, 5, 10240) XSA(t5, 144, 159, b2, 5, 11264) XSA(t0, 128, 143, b3, 5, 12288) XD(0, 1024) \
    XSA(t1, 144, 159, b3, 5, 13312) XSA(t2, 160, 175, b0, 5, 14336) XSA(t3, 176, 191, b0, 5, 15360) \
    XSA(t4, 160, 175, b1, 5, 16384) XD(0, 2048) XSA(t5, 176, 191, b1, 5, 17408) XSA(t0, 160, 175, b2, 5, 18432) \
    XSA(t1, 176, 191, b2, 5, 19456) XSA(t2, 160, 175, b3, 5, 20480) XD(0, 3072) XDADV XSA(t3, 176, 191, b3, 5, 21504) \
    XSA(t4, 192, 207, b0, 5, 22528) XSA(t5, 208, 223, b0, 5, 23552) XSA(t0, 192, 207, b1, 5, 24576) XD(4096, 0) \
    XSA(t1, 208, 223, b1, 5, 25600) XSA(t2, 192, 207, b2, 5, 26624) XSA(t3, 208, 223, b2, 5, 27648) \
    XSA(t4, 192, 207, b3, 5, 28672) XD(4096, 1024) XSA(t5, 208, 223, b3, 5, 29696) XSA(t0, 224, 239, b0, 5, 30720) \
    XSA(t1, 240, 255, b0, 5, 31744) XTA(t2, 224, 239, b1, 5) XD(4096, 2048) XTA(t3, 240, 255, b1, 4) \
    XTA(t4, 224, 239, b2, 3) XTA(t5, 240, 255, b2, 2) XTA(t0, 224, 239, b3, 1) XD(4096, 3072) XTA(t1, 240, 255, b3, 0)

#define XG_K \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XS0(t0, c0, b0, 5, 6144) \
    XS0(t1, c1, b0, 5, 7168) XD(0, 0) XS0(t2, c2, b0, 5, 8192) XS(t3, c0, b1, 5, 9216) XS(t4, c1, b1, 5, 10240) \
    XD(0, 1024) XS(t5, c2, b1, 5, 11264) XT(t0, c0, b2, 5) XT(t1, c1, b2, 4) XD(0, 2048) XT(t2, c2, b2, 3) \
    XT(t3, c0, b3, 2) XT(t4, c1, b3, 1) XD(0, 3072) XDADV XT(t5, c2, b3, 0) XNOP

#define XG_V \
    XRD(t0, 12288) XRD(t1, 13312) XRD(t2, 14336) XRD(t3, 15360) XRD(t4, 16384) XRD(t5, 17408) XS0(t0, c0, b0, 5, 18432) \
    XS0(t1, c1, b0, 5, 19456) XD(4096, 0) XS(t2, c0, b1, 5, 20480) XS(t3, c1, b1, 5, 21504) XS(t4, c0, b2, 5, 22528) \
    XD(4096, 1024) XS(t5, c1, b2, 5, 23552) XT(t0, c0, b3, 5) XT(t1, c1, b3, 4) XD(4096, 2048) XT(t2, c0, b4, 3) \
    XT(t3, c1, b4, 2) XT(t4, c0, b5, 1) XD(4096, 3072) XT(t5, c1, b5, 0) XNOP
#define XG_WD32 \
    XRD(t0, 0) XRD(t1, 1024) XRD(t2, 2048) XRD(t3, 3072) XRD(t4, 4096) XRD(t5, 5120) XSA(t0, 0, 15, b0, 5, 6144) \
    XSA(t1, 16, 31, b0, 5, 7168) XSA(t2, 0, 15, b1, 5, 8192) XD(0, 0) XSA(t3, 16, 31, b1, 5, 9216) \
    XSA(t4, 32, 47, b0, 5, 10240) XSA(t5, 48, 63, b0, 5, 11264) XSA(t0, 32, 47, b1, 5, 12288) XD(0, 1024) \
    XSA(t1, 48, 63, b1, 5, 13312) XSA(t2, 64, 79, b0, 5, 14336) XSA(t3, 80, 95, b0, 5, 15360) \
    XSA(t4, 64, 79, b1, 5, 16384) XD(0, 2048) XSA(t5, 80, 95, b1, 5, 17408) XSA(t0, 96, 111, b0, 5, 18432) \
    XSA(t1, 112, 127, b0, 5, 19456) XSA(t2, 96, 111, b1, 5, 20480) XD(0, 3072) XDADV XSA(t3, 112, 127, b1, 5, 21504) \
    XSA(t4, 128, 143, b0, 5, 22528) XSA(t5, 144, 159, b0, 5, 23552) XSA(t0, 128, 143, b1, 5, 24576) XD(4096, 0) \
    XSA(t1, 144, 159, b1, 5, 25600) XSA(t2, 160, 175, b0, 5, 26624) XSA(t3, 176, 191, b0, 5, 27648) \
    XSA(t4, 160, 175, b1, 5, 28672) XD(4096, 1024) XSA(t5, 176, 191, b1, 5, 29696) XSA(t0, 192, 207, b0, 5, 30720) \
    XSA(t1, 208, 223, b0, 5, 31744) XTA(t2, 192, 207, b1, 5) XD(4096, 2048) XTA(t3, 208, 223, b1, 4) \
    XTA(t4, 224, 239, b0, 3) XTA(t5, 240, 255, b0, 2) XTA(t0, 224, 239, b1, 1) XD(4096, 3072) XTA(t1, 240, 255, b1, 0)

#define XTMP_OUT [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [so] "+s"(nx.so)
#define XDMA_IN [ldsn] "s"(nx.ldsn), [srd] "s"(nx.srd), [voff] "v"(voff)

// ---- the head loop of ONE text cross-attention sub-layer on the wave's 32 tokens (shared by the cross-attention kernel and the block
// kernel below): groups sg0 .. sg0 + 39 of the stream ----------------------------------------------------------------------------------
struct XNext { uint4_t srd; unsigned so, ldsn; };         // the group XRING - 1 = 3 ahead: its source and its ring slot
template <int TR, class GS>
UAV_DEVINL void xattn_heads(const int sg0, half8_t (&xn)[32], GS&& group_sync, XNext& nx, const unsigned voff, const int hi, const int lk,
                            const float scale_log2, unsigned long long (&ts)[12], const bool stamp) {
#pragma unroll 1
    for (int h = 0; h < XHEADS; ++h) {
        half8_t t0, t1, t2, t3, t4, t5;
        const int sg = sg0 + h * XGPH;                // first group of this head in the stream
        if (TR && stamp && h == 1) ts[3] = __builtin_amdgcn_s_memtime();    // head 1 is stamped phase by phase (head 0 carries the cold start)
        // Q_h^T [64 ch][32 tokens] = Wq_h . Xn^T
        float16_t q0, q1;
        {
            const unsigned st = group_sync(sg);
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
            const unsigned st = group_sync(sg + j);
            asm volatile(XG_WQ : [q0] "+v"(q0), [q1] "+v"(q1), XTMP_OUT
                         : [st] "v"(st), [b0] "v"(xn[16 * j + 0]), [b1] "v"(xn[16 * j + 1]), [b2] "v"(xn[16 * j + 2]), [b3] "v"(xn[16 * j + 3]),
                           [b4] "v"(xn[16 * j + 4]), [b5] "v"(xn[16 * j + 5]), [b6] "v"(xn[16 * j + 6]), [b7] "v"(xn[16 * j + 7]),
                           [b8] "v"(xn[16 * j + 8]), [b9] "v"(xn[16 * j + 9]), [b10] "v"(xn[16 * j + 10]), [b11] "v"(xn[16 * j + 11]),
                           [b12] "v"(xn[16 * j + 12]), [b13] "v"(xn[16 * j + 13]), [b14] "v"(xn[16 * j + 14]), [b15] "v"(xn[16 * j + 15]), XDMA_IN
                         : "memory", "scc");
        }
        if (TR && stamp && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[4] = __builtin_amdgcn_s_memtime(); }     // Q GEMM (64 MFMA)
        half8_t qf[4];                                      // Q rounded to fp16 like the stored q of the unfused chain
#pragma unroll
        for (int e = 0; e < 8; ++e) { qf[0][e] = (half_t)q0[e]; qf[1][e] = (half_t)q0[8 + e]; qf[2][e] = (half_t)q1[e]; qf[3][e] = (half_t)q1[8 + e]; }
        // S^T [96 keys][32 tokens] = K_h . Q^T
        float16_t sacc[3];
        const unsigned stkv = group_sync(sg + 2);
        asm volatile(XG_K : [c0] "=&v"(sacc[0]), [c1] "=&v"(sacc[1]), [c2] "=&v"(sacc[2]), XTMP_OUT
                     : [st] "v"(stkv), [b0] "v"(qf[0]), [b1] "v"(qf[1]), [b2] "v"(qf[2]), [b3] "v"(qf[3]), XDMA_IN : "memory", "scc");
        if (TR && stamp && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[5] = __builtin_amdgcn_s_memtime(); }     // S = K Q (12 MFMA)
        // softmax over the keys: this lane holds keys 32 t + (r & 3) + 8 (r >> 2) + 4 hi, lane ^ 32 the others
        float mx = -INFINITY;
        int lk_ = lk;
        asm volatile("" : "+s"(lk_));                        // (re-read per head: hipcc otherwise hoists 48 key compares out of both head loops and
                                                            //  pays for their 96 mask registers with spills)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (32 * (t + 1) <= lk_) {                      // wave-uniform: a key tile without padding needs no mask
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float s = sacc[t][r] * scale_log2; sacc[t][r] = s; mx = fmaxf(mx, s); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float s = sacc[t][r] * scale_log2;
                    s = key < lk_ ? s : -INFINITY;
                    sacc[t][r] = s; mx = fmaxf(mx, s);
                }
            }
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
        if (TR && stamp && h == 1) ts[6] = __builtin_amdgcn_s_memtime();                                                          // softmax
        // O^T [64 ch][32 tokens] = V_h^T . P^T (same LDS slot, fragments 12 .. 23)
        float16_t o0, o1;
        asm volatile(XG_V : [c0] "=&v"(o0), [c1] "=&v"(o1), XTMP_OUT
                     : [st] "v"(stkv), [b0] "v"(pf[0]), [b1] "v"(pf[1]), [b2] "v"(pf[2]), [b3] "v"(pf[3]), [b4] "v"(pf[4]), [b5] "v"(pf[5]), XDMA_IN
                     : "memory", "scc");
        if (TR && stamp && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[7] = __builtin_amdgcn_s_memtime(); }     // O = V P (12 MFMA)
        half8_t of[4];                                      // O / l rounded to fp16 like the stored attention output
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            of[0][e] = (half_t)(o0[e] * inv); of[1][e] = (half_t)(o0[8 + e] * inv);
            of[2][e] = (half_t)(o1[e] * inv); of[3][e] = (half_t)(o1[8 + e] * inv);
        }
        // acc [512 ch][32 tokens] += Wout[:, head h] . O^T (named accumulators: tiles 0 .. 7, then 8 .. 15)
        {
            const unsigned st = group_sync(sg + 3);
            asm volatile(XG_WO0 : XTMP_OUT : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), [b2] "v"(of[2]), [b3] "v"(of[3]), XDMA_IN
                         : "memory", "scc", XACC_CLOBBERS);
        }
        {
            const unsigned st = group_sync(sg + 4);
            asm volatile(XG_WO1 : XTMP_OUT : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), [b2] "v"(of[2]), [b3] "v"(of[3]), XDMA_IN
                         : "memory", "scc", XACC_CLOBBERS);
        }
        if (TR && stamp && h == 1) { asm volatile("s_nop 15\ns_nop 15" ::: "memory"); ts[8] = __builtin_amdgcn_s_memtime(); }     // acc += Wout O (64 MFMA)
    }
}

// ---- LayerNorm of the NEXT sub-layer on the rows the accumulators hold (ltab: LDS address of its gamma | beta | bias tables) -------------
UAV_DEVINL void mid_layernorm(half8_t (&xn)[32], const unsigned ltab, const float eps, const int hi) {
        // ---- the NEXT sub-layer of the block on the same tile: its input is what the accumulators hold (the first sub-layer's output —
        // fp32, exactly the rows the four-launch chain would have written and read back), so its LayerNorm runs on them in place: two
        // passes like layernorm_kernel, new operand fragments over the old, + its output bias.  One prologue and one epilogue for two
        // sub-layers, and the stream between them never touches HBM. -----------------------------------------------------------------
        asm volatile("s_nop 15\ns_nop 15" ::: "memory");    // the last MFMAs of the head loop may still be in flight and the compiler cannot see them
        float sm = 0.f;
        static_for<256>([&](auto N) { sm += acc_get<N>(); });
        sm += swap32(sm);
        const float mean2 = sm * (1.0f / XC);
        float sq = 0.f;
        static_for<256>([&](auto N) { const float d = acc_get<N>() - mean2; sq += d * d; });
        sq += swap32(sq);
        const float rstd2 = rsqrtf(sq * (1.0f / XC) + eps);
        static_for<16>([&](auto J) {
            constexpr int j = J;
            static_for<4>([&](auto Q) {
                constexpr int q = Q;
                const unsigned ta = ltab + (32 * j + 8 * q + 4 * hi) * 4;
                const float4_t g = lds_f4(ta), be = lds_f4(ta + 2048), bo = lds_f4(ta + 4096);
                static_for<4>([&](auto I) {
                    constexpr int i = I;
                    const float v = acc_get<16 * j + 4 * q + i>();
                    xn[2 * j + (q >> 1)][4 * (q & 1) + i] = (half_t)((v - mean2) * rstd2 * g[i] + be[i]);
                    acc_set<16 * j + 4 * q + i>(v + bo[i]);
                });
            });
        });
}

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
    s1 += swap32(s1); s2 += swap32(s2);
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
};

// NX = 0: the temporal sub-layer alone.  NX = 2: attn1 -> attn2 -> attn_temporal of one BasicTransformerBlock (only_cross_attention) in ONE
// launch on the temporal tiling — a workgroup's 16 pixels x 8 frames lie inside one batch entry, which is all the cross-attention head
// loop asks of its 32 tokens —: the stream is read once and written once for three sub-layers, the rows between them stay in the
// accumulators and every LayerNorm but the first runs on them in registers.
template <int NX>
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
    constexpr int SG_T = NX * XNG;                         // first group of the temporal sub-layer in the stream
    auto next_of = [&](int s) -> XNext {
        XNext n;
        n.ldsn = lds0 + (unsigned)((s & (XRING - 1)) * XGROUP + wave * XPPW * XFRAG);
        if (NX > 0 && s < SG_T) {                          // a cross-attention sub-layer: head r / 5, group j = r % 5 (0, 1: W_q; 2: K | V; 3, 4: W_out)
            const int u = s >= XNG ? 1 : 0, r = s - u * XNG;
            const int h = r / XGPH, j = r - h * XGPH;
            const XattnSub& S = p.xs[u];
            if (j < 2) { n.srd = make_srd(S.wq, XHEADS * 2 * XGROUP); n.so = (unsigned)((h * 2 + j) * XGROUP); }
            else if (j == 2) { n.srd = make_srd(S.kv + (long long)bb * XHEADS * XGROUP, XHEADS * XGROUP); n.so = (unsigned)(h * XGROUP); }
            else { n.srd = make_srd(S.wo, XHEADS * 2 * XGROUP); n.so = (unsigned)((h * 2 + (j - 3)) * XGROUP); }
            return n;
        }
        const int st = s - SG_T;                           // temporal: st = 8 h + j: j 0, 1: W_q; 2, 3: W_k; 4, 5: W_v; 6, 7: W_out
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
        if (p.ln_out) {
            *(lds_f4wptr_t)(size_t)(lds0 + TTAB_LN3 + tid * 16) = ((const float4_t*)p.ln_gamma)[tid];
            *(lds_f4wptr_t)(size_t)(lds0 + TTAB_LN3 + 2048 + tid * 16) = ((const float4_t*)p.ln_beta)[tid];
        }
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
    s1 += swap32(s1); s2 += swap32(s2);
    const float m1 = s1 * (1.0f / XC);
    const float mean = c0 + m1;
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / XC) - m1 * m1, 0.f) + (NX > 0 ? p.xs[0].eps : p.eps));      // (the tables at XTAB are the first sub-layer's)
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
                acc_set<16 * j + 4 * q + i>(v[i] + bo[i]);
            });
        });
        if (j & 1) __builtin_amdgcn_sched_barrier(0);       // batches of 8 loads (this prologue carries the row arithmetic of the frame-strided tile on top)
    });

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
        mx = fmaxf(mx, swap32(mx));
        float ps = 0.f;
        half8_t pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f((sacc[r] - mx) * 1.44269504088896341f);
            ps += e;
            pf[r >> 3][r & 7] = (half_t)e;
        }
        ps += swap32(ps);
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
        xattn_heads<0>(0, xn, group_sync, nx, voff, hi, p.lk, p.xscale_log2, ts_, false);
        mid_layernorm(xn, lds0 + XTAB + XTABS, p.xs[1].eps, hi);
        xattn_heads<0>(XNG, xn, group_sync, nx, voff, hi, p.lk, p.xscale_log2, ts_, false);
        // (a second fragment array: hipcc gives the temporal loop's fragments other registers than the cross loops', and moving one set
        //  onto the other through the full register file went through scratch — 33 spilled fragments)
        half8_t xt[32];
        mid_layernorm(xt, lds0 + TT_LN, p.eps, hi);
        temporal_heads(xt);
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
            sm += swap32(sm);
            const float mean3 = sm * (1.0f / XC);
            float sq = 0.f;
            static_for<256>([&](auto N) { const float d = acc_get<N>() - mean3; sq += d * d; });
            sq += swap32(sq);
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
constexpr int FSLICES = 64;                    // hidden channels in slices of 32
constexpr int FGPS = 3;                        // groups per slice: W_up value | gate rows (2), W_down columns (1)
constexpr int FNG = FSLICES * FGPS;
constexpr int FINNER = FSLICES * 32;
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
    s1 += swap32(s1); s2 += swap32(s2);
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

    // ---- slices ------------------------------------------------------------------------------------------------------------------
    {
        int lane2;                                          // (fresh lane id: see the kernels above)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
        const int hi2 = lane2 >> 5;
        lane16 = (unsigned)lane2 * 16;
#pragma unroll 1
        for (int c = 0; c < FSLICES; ++c) {
            half8_t t0, t1, t2, t3, t4, t5;
            const int sg = c * FGPS;
            // value^T (q0), gate^T (q1) [32 ch][32 tokens] = W_up[value / gate rows of the slice] . Xn^T
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
            // GEGLU on the D layout: register r <-> hidden channel 32 c + (r & 3) + 8 (r >> 2) + 4 hi; fp16 = the B fragments of the down
            // step (k-step r >> 3)
            half8_t of[2];
            const unsigned ub = lds0 + FTAB_UP + (32 * c + 4 * hi2) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4_t bv = lds_f4(ub + 32 * q), bg = lds_f4(ub + FINNER * 4 + 32 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    of[q >> 1][4 * (q & 1) + i] = (half_t)((q0[r] + bv[i]) * uav_gelu_erf(q1[r] + bg[i]));
                }
            }
            // acc [512 ch][32 tokens] += W_down[:, slice c] . H^T
            {
                const unsigned st = group_sync(sg + 2);
                asm volatile(XG_WD32 : XTMP_OUT : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), XDMA_IN : "memory", "scc", XACC_CLOBBERS);
            }
        }
    }
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

namespace {
int tattn_launch(const float* x, float* out, const uav_xattn_params* xs, int32_t n_xs, int32_t lk, float xscale, const uav_tattn_params* q,
                 int32_t n_batch, int32_t t_len, int64_t hw, int32_t channels, int32_t heads, float scale, void* stream) {
    if (!x || !out || !q || !q->ln_gamma || !q->ln_beta || !q->wq_packed || !q->wk_packed || !q->wv_packed || !q->wo_packed || !q->out_bias ||
        !q->rel_bias || !q->rope_cos || !q->rope_sin)
        return UAV_EINVAL;
    if (channels != XC || heads != XHEADS || t_len != TT || q->rot_dim != 32) return UAV_ESHAPE;
    if (n_batch <= 0 || hw <= 0 || (hw % 16) || (long long)n_batch * (hw / 16) >= (1ll << 31)) return UAV_ESHAPE;
    if (n_xs != 0 && n_xs != 2) return UAV_ESHAPE;
    if (n_xs && (!xs || lk <= 0 || lk > 96)) return UAV_ESHAPE;
    if (((size_t)x | (size_t)out) & 15) return UAV_EALIGN;
    TattnArgs a{x, out, q->ln_gamma, q->ln_beta, q->out_bias, q->ln_eps, (const char*)q->wq_packed, (const char*)q->wk_packed,
                (const char*)q->wv_packed, (const char*)q->wo_packed, q->rel_bias, q->rope_cos, q->rope_sin, n_batch, (long long)hw, scale,
                {}, lk, xscale * 1.44269504088896341f, (half_t*)q->next_ln_out, q->next_ln_gamma, q->next_ln_beta, q->next_ln_eps};
    if (q->next_ln_out && (!q->next_ln_gamma || !q->next_ln_beta || ((size_t)q->next_ln_out & 15))) return UAV_EINVAL;
    for (int i = 0; i < n_xs; ++i) {
        const uav_xattn_params& c = xs[i];
        if (!c.ln_gamma || !c.ln_beta || !c.wq_packed || !c.kv_packed || !c.wo_packed || !c.out_bias) return UAV_EINVAL;
        a.xs[i] = XattnSub{c.ln_gamma, c.ln_beta, c.out_bias, (const char*)c.wq_packed, (const char*)c.kv_packed, (const char*)c.wo_packed, c.ln_eps};
    }
    const dim3 grid((unsigned)(n_batch * (hw / 16)));
    if (n_xs) {
        static UavDynLds lds2;
        if (int rc = uav_set_dyn_lds(lds2, (const void*)tattn_sublayer_kernel<2>, TSMEM)) return rc;
        hipLaunchKernelGGL(tattn_sublayer_kernel<2>, grid, dim3(256), TSMEM, (hipStream_t)stream, a);
    } else {
        static UavDynLds lds0;
        if (int rc = uav_set_dyn_lds(lds0, (const void*)tattn_sublayer_kernel<0>, TSMEM)) return rc;
        hipLaunchKernelGGL(tattn_sublayer_kernel<0>, grid, dim3(256), TSMEM, (hipStream_t)stream, a);
    }
    return uav_launch_status();
}
}  // namespace

extern "C" int uav_tattn_sublayer_f32(const float* x, float* out, const uav_tattn_params* q, int32_t n_batch, int32_t t_len, int64_t hw,
                                      int32_t channels, int32_t heads, float scale, void* stream) {
    return tattn_launch(x, out, nullptr, 0, 0, 0.f, q, n_batch, t_len, hw, channels, heads, scale, stream);
}

extern "C" int uav_block_attn_sublayers_f32(const float* x, float* out, const uav_xattn_params* cross, int32_t lk, float cross_scale,
                                            const uav_tattn_params* temporal, int32_t n_batch, int32_t t_len, int64_t hw, int32_t channels,
                                            int32_t heads, float temporal_scale, void* stream) {
    return tattn_launch(x, out, cross, 2, lk, cross_scale, temporal, n_batch, t_len, hw, channels, heads, temporal_scale, stream);
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
