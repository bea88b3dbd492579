// Shared pieces of the fused transformer sub-layer kernels (xattn_fused.hip, tattn_fused.hip, tattn_block_fused.hip): the fragment-group
// asm walks, the LDS-DMA ring helpers, the named-accumulator access, the cross-attention head loop, the in-register LayerNorm and the
// feed-forward slice loop.  Design notes: the header comment of xattn_fused.hip.
#pragma once
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
// The two halves of the wave (lane, lane ^ 32 = the two halves of a token's row / key set) combined.  UAV_HALF_REDUCE_PERMLANE (the temporal /
// block kernels, tattn_kernel.h): gfx950's v_permlane32_swap — with both operands = v it leaves the lower half's value in every lane of one
// register and the upper half's in the other, so sum / max of the two is the reduction in ALL lanes, without the lane-address register of
// __shfl_xor's ds_bpermute (one VGPR live across every head loop: in the whole-block kernel hipcc parked it in a0, a NAMED accumulator — the
// build audit caught it) and without a trip through the LDS crossbar.  (As inline asm: the builtin, fed the same value twice, was folded to
// a + a by this hipcc.)  The cross-attention and feed-forward kernels keep the shuffle: with the swap their register allocation, which sits
// at the 256-VGPR limit, came out with 2 (then 49) spilled registers.  Same bits either way (a + b and max are commutative).
#ifdef UAV_HALF_REDUCE_PERMLANE
UAV_DEVINL void half_swap(float& a, float& b) { asm volatile("s_nop 1\nv_permlane32_swap_b32 %0, %1\ns_nop 1" : "+v"(a), "+v"(b)); }
UAV_DEVINL float half_sum(float v) { float a = v, b = v; half_swap(a, b); return a + b; }
UAV_DEVINL float half_max(float v) { float a = v, b = v; half_swap(a, b); return fmaxf(a, b); }
#else
UAV_DEVINL float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }
UAV_DEVINL float half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
#endif
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
        mx = half_max(mx);
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
        ps = half_sum(ps);
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
        sm = half_sum(sm);
        const float mean2 = sm * (1.0f / XC);
        float sq = 0.f;
        static_for<256>([&](auto N) { const float d = acc_get<N>() - mean2; sq += d * d; });
        sq = half_sum(sq);
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

// ---- the feed-forward sub-layer on the wave's 32 tokens (ff_sublayer_kernel, and behind the three attention sub-layers in the block
// kernel): 64 slices of 32 hidden channels, groups sg0 .. sg0 + 191 of the stream; ub0: LDS address of b_up (value 0 .. 2047 | gate) -----------
constexpr int FSLICES = 64;                    // hidden channels in slices of 32
constexpr int FGPS = 3;                        // groups per slice: W_up value | gate rows (2), W_down columns (1)
constexpr int FNG = FSLICES * FGPS;
constexpr int FINNER = FSLICES * 32;

// uav_gelu_erf (uav_common.h: Abramowitz & Stegun 7.1.26 on z = x / sqrt 2) with the constants folded onto x — t = 1 / (1 + (p / sqrt 2) |x|),
// exp(-z^2) = exp2(-(log2 e / 2) x^2), x (0.5 + 0.5 erf) — five VALU operations less per value (the slice's GEGLU is ~20 % of the
// feed-forward phase and does not overlap with its MFMAs); the same polynomial, fp32 rounding apart.
UAV_DEVINL float ff_gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * x * x);
    const float hr = fmaf(-0.5f * poly * t, e, 0.5f);      // 0.5 erf(|z|)
    return fmaf(x, 0.5f, ax * hr);                          // x (0.5 + 0.5 sign(x) erf(|z|)) = 0.5 x + |x| * 0.5 erf(|z|)
}

template <class GS>
UAV_DEVINL void ff_slices(const int sg0, half8_t (&xn)[32], GS&& group_sync, XNext& nx, const unsigned voff, unsigned& lane16, const unsigned ub0) {
    int lane2;                                              // (fresh lane id: see the kernels above)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\nv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
    const int hi2 = lane2 >> 5;
    lane16 = (unsigned)lane2 * 16;
#pragma unroll 1
    for (int c = 0; c < FSLICES; ++c) {
        half8_t t0, t1, t2, t3, t4, t5;
        const int sg = sg0 + c * FGPS;
        // value^T (q0), gate^T (q1) [32 ch][32 tokens] = W_up[value / gate rows of the slice] . Xn^T
        // (the accumulators start from the biases b_v | b_g of the slice's channels: register r <-> hidden channel 32 c + (r & 3) + 8 (r >> 2)
        //  + 4 hi — 32 additions less per slice than adding them behind the GEMM)
        float16_t q0, q1;
        {
            const unsigned ub = ub0 + (32 * c + 4 * hi2) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4_t bv = lds_f4(ub + 32 * q), bg = lds_f4(ub + FINNER * 4 + 32 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) { q0[4 * q + i] = bv[i]; q1[4 * q + i] = bg[i]; }
            }
        }
        {
            const unsigned st = group_sync(sg);
            asm volatile(XG_WQ : [q0] "+v"(q0), [q1] "+v"(q1), XTMP_OUT
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
#pragma unroll
        for (int r = 0; r < 16; ++r) of[r >> 3][r & 7] = (half_t)(q0[r] * ff_gelu_erf(q1[r]));
        // acc [512 ch][32 tokens] += W_down[:, slice c] . H^T
        {
            const unsigned st = group_sync(sg + 2);
            asm volatile(XG_WD32 : XTMP_OUT : [st] "v"(st), [b0] "v"(of[0]), [b1] "v"(of[1]), XDMA_IN : "memory", "scc", XACC_CLOBBERS);
        }
    }
}

}  // namespace
