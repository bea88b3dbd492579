// K1/K2 — implicit-GEMM convolution / linear: parameter checks, kernel selection and the C ABI (uav_conv_gemm_f16 and its host-only
// queries).  The kernels live in one translation unit per family: see conv_common.h.
#include "conv_common.h"

namespace {
struct ConvEnv { int korder, tile_order, force_tile, dbg, persist, dmav, sk, sk_maxk, w4, w4_mink; };
const ConvEnv& conv_env() {
    // Environment switches (development A/B only) are read once through a thread-safe magic static.
    static const ConvEnv env = [] {
        auto geti = [](const char* k, int d) { const char* e = getenv(k); return e ? atoi(e) : d; };
#ifdef UAV_DEV_KERNELS
        {   // only the loops that still have instances: an unknown value used to fall through to the round-1 kernel silently
            const int v = geti("UAV_CONV_DMAV", 6);
            if (v != 1 && v != 6) { fprintf(stderr, "[uav] UAV_CONV_DMAV=%d has no kernel instance (1: round 2-3 loop, 6: rotated k-step); using 6\n", v); setenv("UAV_CONV_DMAV", "6", 1); }
        }
        const int dbg = geti("UAV_CONV_DBG", 0), persist = geti("UAV_CONV_PERSIST", 0), dmav = geti("UAV_CONV_DMAV", 6);
        const int sk = geti("UAV_CONV_SK", 0), sk_maxk = geti("UAV_CONV_SK_MAXK", 1024);
#else
        // product library: the ablation builds, the round 2-3 loop, the persistent round-1 walk and the short-K kernel are not in it
        // (csrc/conv_gemm_dev.hip, -DUAV_DEV_KERNELS); their switches are not read
        const int dbg = 0, persist = 0, dmav = 6, sk = 0, sk_maxk = 0;
#endif
        // W4 for K = taps x C_in >= MINK.  Same-box clip A/Bs: run 10 of round 5 (lane-per-row epilogues) 1.000 (8-wave everywhere) -> 1.057 (W4
        // everywhere) -> 1.070 (from K = 1024); run 25 (fp32 epilogues through LDS) 1.141 (from 1024) -> 1.142 (768) -> 1.152 (512) -> 1.154 (256)
        // -> 1.155 (everywhere): 0
        return ConvEnv{geti("UAV_CONV_KORDER", 1), geti("UAV_CONV_TILE_ORDER", 1), geti("UAV_CONV_TILE", 0), dbg, persist, dmav, sk, sk_maxk,
                       geti("UAV_CONV_W4", 1), geti("UAV_CONV_W4_MINK", 0)};
    }();
    return env;
}
// tile selection: the 256x256 kernel needs n_pad % 256 == 0 and enough tiles to fill 256 CUs
bool conv_uses_big_tile(const uav_conv_params* q) {
    const bool small = (q->c1 == 8 && q->c2 == 0);
    const long long M = (long long)q->n_img * q->ho * q->wo;
    const long long grid256 = ((M + LM - 1) / LM) * (q->n_pad / LN);
    const int force_tile = conv_env().force_tile;
    return !small && (q->n_pad % LN == 0) && (force_tile >= 256 || (force_tile != 128 && grid256 >= 224));
}
// n / d for n < 2^31 as umulhi(n, mul) >> sh (d >= 2: s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1, sh = s - 1; d == 1: sh = 32 = "no division")
void conv_magic(unsigned d, unsigned* mul, unsigned* sh) {
    if (d <= 1) { *mul = 0; *sh = 32; return; }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *mul = (unsigned)(((1ull << (31 + s)) / d) + 1ull);
    *sh = s - 1;
}
// Four-wave kernel (conv_gemm256w_kernel): every launch of the big-tile class whose taps fit a 32-bit validity mask and whose
// sources fit a 32-bit buffer range; LayerNorm-fold instances, the nearest-2x gather and the A/B switches stay with the 8-wave kernels.
bool conv_uses_w4(const uav_conv_params* q) {
    const ConvEnv& env = conv_env();
    if (!env.w4 || env.dbg || env.persist || !env.korder || env.dmav != 6 || (q->flags & (UAV_CONV_PERSISTENT | UAV_CONV_NO_W4))) return false;
    if (q->upsample || q->ln_raw_out || q->ln_stat_in || q->kt * q->kh * q->kw > 32 || q->kt > 8 || q->kh > 8 || q->kw > 8) return false;
    if ((long long)q->kt * q->kh * q->kw * (q->c1 + q->c2) < env.w4_mink) return false;      // A/B: short K stays with the 8-wave kernel
    const unsigned long long px = (unsigned long long)q->n_img * q->hi * q->wi;
    const unsigned long long a2px = q->a2_images ? px / 2 : px;
    const unsigned long long halo = 2ull * (((unsigned long long)q->pad_t * q->hi + q->pad_h) * q->wi + q->pad_w) + 16;   // pixels of displacement range
    if ((px + halo) * q->c1 * 2 >= 0xfffffff0ull || (a2px + halo) * q->c2 * 2 >= 0xfffffff0ull) return false;
    return conv_uses_big_tile(q);
}
// UAV_CONV_OUT_HILO (block tails written as the hi | lo fp16 operand pair of their 1x1 consumer): only the row-coalesced fp32
// epilogue of the four-wave kernel stores that form, so the launch must run there and EVERY wave tile must take it (conv_co_kind).
bool conv_hilo_ok(const uav_conv_params* q) {
    if (!(q->flags & UAV_CONV_OUT_F32) || !conv_uses_w4(q)) return false;
    const long long M = (long long)q->n_img * q->ho * q->wo;
    const bool rf32 = q->flags & UAV_CONV_RES_F32;
    if ((q->flags & (UAV_CONV_GELU | UAV_CONV_QUICK_GELU | UAV_CONV_GEGLU)) || !q->bias || (M % 64) || (q->n % 128) || (q->out_stride & 3) ||
        q->out_stride < 2 * q->n || !q->residual || !rf32 || (q->res_stride & 3) || q->rowbias)
        return false;                        // block tails are residual sums on the fp32 stream: the one epilogue form instantiated
    return !q->gn_partials;                  // (the statistics instances do not carry the pair store)
}
#ifdef UAV_DEV_KERNELS
// Short-K kernel: 1x1 / stride 1 launches of the big-tile class with K <= UAV_CONV_SK_MAXK and whole 256-column tiles.
bool conv_uses_sk(const uav_conv_params* q) {
    const ConvEnv& env = conv_env();
    if (!env.sk || env.dbg || env.persist || !env.korder || (q->flags & (UAV_CONV_PERSISTENT | UAV_CONV_NO_SHORTK))) return false;
    if (q->kt != 1 || q->kh != 1 || q->kw != 1 || q->stride != 1 || q->upsample || q->pad_t || q->pad_h || q->pad_w) return false;
    if (q->ho != q->hi || q->wo != q->wi || q->out_map_w || q->a2_center_tap) return false;
    if (q->ln_raw_out || q->ln_stat_in) return false;
    if (q->n != q->n_pad || (q->n_pad % 256) || q->k_pad != q->c1 + q->c2 || q->k_pad > env.sk_maxk) return false;
    return conv_uses_big_tile(q);
}
#endif
// Fused GroupNorm statistics are produced by the fast epilogues of the 256x256 kernel only: every wave tile (64 rows x
// 128 channels) must lie inside M x N and qualify for a fast path, and a group must not straddle wave tiles.
int conv_gn_cpg_log2(const uav_conv_params* q) {
    // chunks are runs of consecutive output rows — or, for strided output rows (out_map), placed by gn_chunk_* in a shared workspace
    if (q->gn_groups <= 0 || (q->n % q->gn_groups) || (q->out_map_w && q->gn_chunk_cpi <= 0)) return -1;
    const int cpg = q->n / q->gn_groups;
    int cl = -1;
    for (int k = 2; k <= 7; ++k) if (cpg == (1 << k)) cl = k;
    if (cl < 0) return -1;
    const long long M = (long long)q->n_img * q->ho * q->wo;
    if (!conv_uses_big_tile(q) || (M % 64) || (q->n % 128)) return -1;        // (the short-K kernel has the same 64 x 128 wave tiles)
    if (q->flags & (UAV_CONV_GEGLU | UAV_CONV_GELU | UAV_CONV_QUICK_GELU)) return -1;
    const bool of32 = q->flags & UAV_CONV_OUT_F32, rf32 = q->flags & UAV_CONV_RES_F32;
    if (of32) {
        if (!q->bias || (q->out_stride & 3) || (q->residual && (!rf32 || (q->res_stride & 3)))) return -1;
        if (q->rowbias && (q->residual || (q->rows_per_batch % 64))) return -1;
    } else if (rf32) {                                          // fp32 residual, fp16 result
        if (!q->bias || q->rowbias || (q->out_stride & 7) || (q->res_stride & 3)) return -1;
    } else {
        if ((q->out_stride & 7) || (q->residual && (q->res_stride & 7))) return -1;
        if (q->rowbias && (q->rows_per_batch % 64)) return -1;
    }
    return cl;
}
// LayerNorm fold (producer: ln_raw_out + ln_stat_out; consumer: ln_stat_in + ln_colsum): only launches whose EVERY wave tile takes
// the staged fast epilogue of the production 256x256 kernel.
bool conv_ln_ok(const uav_conv_params* q) {
#ifndef UAV_DEV_KERNELS
    (void)q;
    return false;                              // the LayerNorm-fold instances (measured slower and outside the parity bar) live in the development build
#else
    const long long M = (long long)q->n_img * q->ho * q->wo;
    const ConvEnv& env = conv_env();
    if (!conv_uses_big_tile(q) || (M % 64) || (q->n % 128) || q->out_map_w || q->gn_partials || q->rowbias || !q->bias) return false;
    if (env.dbg || env.persist || (env.dmav != 1 && env.dmav != 6) ||
        (q->flags & (UAV_CONV_PERSISTENT | UAV_CONV_GELU | UAV_CONV_QUICK_GELU)))
        return false;                                      // (the LayerNorm-fold instances themselves run the V = 1 loop)
    const bool of32 = q->flags & UAV_CONV_OUT_F32, rf32 = q->flags & UAV_CONV_RES_F32;
    if (q->ln_raw_out) {                                   // producer: fp32 result (+ fp32 residual), rows of n = out_stride values
        if (!q->ln_stat_out || !of32 || (q->flags & UAV_CONV_GEGLU) || (q->out_stride & 3) || q->out_stride != q->n ||
            (q->residual && (!rf32 || (q->res_stride & 3))))
            return false;
    }
    if (q->ln_stat_in) {                                   // consumer: fp16 result, no residual
        if (!q->ln_colsum || of32 || q->residual || (q->out_stride & 7) || q->ln_chunks <= 0 || q->ln_n <= 0 || q->ln_chunks * 128 != q->ln_n ||
            q->ln_n != q->c1 || q->c2 || q->kt * q->kh * q->kw != 1 || ((q->flags & UAV_CONV_GEGLU) && (q->n % 64)))
            return false;
    }
    return true;
#endif
}
}  // namespace

extern "C" int uav_conv_gemm_ln_ok(const uav_conv_params* q) {
    if (!q || q->n_pad <= 0 || (!q->ln_raw_out && !q->ln_stat_in)) return 0;
    return conv_ln_ok(q) ? 1 : 0;
}

extern "C" int uav_conv_gemm_hilo_ok(const uav_conv_params* q) {
    if (!q || q->n_pad <= 0) return 0;
    return conv_hilo_ok(q) ? 1 : 0;
}

extern "C" int uav_conv_gemm_gn_chunk_rows(const uav_conv_params* q) {
    if (!q || q->n_pad <= 0 || q->gn_groups <= 0) return 0;
    return conv_gn_cpg_log2(q) >= 0 ? 64 : 0;
}

extern "C" int uav_conv_gemm_f16(const uav_conv_params* q, void* stream) {
    if (!q || !q->a1 || !q->w || !q->out || !q->zero_page) return UAV_EINVAL;
    const bool small = (q->c1 == 8 && q->c2 == 0);
    if (!small && ((q->c1 % 64) || (q->c2 % 64) || q->c1 <= 0 || q->c2 < 0)) return UAV_ESHAPE;
    if (q->c2 > 0 && !q->a2) return UAV_EINVAL;
    if ((q->n_pad % BN) || (q->k_pad % BK) || q->n <= 0 || q->n > q->n_pad) return UAV_ESHAPE;
    const int ntaps = q->kt * q->kh * q->kw;
    const int cin = q->c1 + q->c2;
    if (ntaps <= 0 || ntaps > 27 * 4) return UAV_ESHAPE;
    if ((long long)ntaps * cin > q->k_pad) return UAV_ESHAPE;
    if (!small && (long long)ntaps * cin != q->k_pad) return UAV_ESHAPE;   // cin%64==0 => exact
    if (q->n % 4) return UAV_ESHAPE;
    if (q->out_stride % 4 || (q->residual && (q->res_stride % 4))) return UAV_EALIGN;
    if (q->rowbias && (q->rows_per_batch <= 0 || (q->rowbias_stride % 4))) return UAV_ESHAPE;
    if ((q->flags & UAV_CONV_GEGLU) && ((q->flags & UAV_CONV_OUT_F32) || q->residual || q->rowbias || (q->n % 64)))
        return UAV_ESHAPE;
    if ((q->flags & UAV_CONV_RES_F32) && !q->residual) return UAV_EINVAL;
    if ((q->flags & UAV_CONV_OUT_HILO) && !conv_hilo_ok(q)) return UAV_ESHAPE;      // ask uav_conv_gemm_hilo_ok() first
    if ((q->flags & (UAV_CONV_GELU | UAV_CONV_QUICK_GELU)) && (q->flags & UAV_CONV_GEGLU)) return UAV_ESHAPE;
    if (q->upsample && (q->stride != 1 || q->ho != 2 * q->hi || q->wo != 2 * q->wi)) return UAV_ESHAPE;
    if (q->t_len <= 0 || q->n_img % q->t_len) return UAV_ESHAPE;
    if (q->ho >= 65536 || q->wo >= 65536) return UAV_ESHAPE;
    ConvArgs a;
    a.a1 = (const char*)q->a1; a.a2 = (const char*)q->a2; a.c1 = q->c1; a.c2 = q->c2;
    a.w = (const char*)q->w; a.bias = q->bias; a.rowbias = q->rowbias;
    a.rows_per_batch = q->rows_per_batch; a.rowbias_stride = q->rowbias_stride;
    a.residual = (const char*)q->residual; a.res_stride = q->res_stride;
    a.out = (char*)q->out; a.out_stride = q->out_stride;
    a.n_img = q->n_img; a.t_len = q->t_len; a.hi = q->hi; a.wi = q->wi; a.ho = q->ho; a.wo = q->wo;
    a.kt = q->kt; a.kh = q->kh; a.kw = q->kw; a.stride = q->stride;
    a.pad_t = q->pad_t; a.pad_h = q->pad_h; a.pad_w = q->pad_w; a.upsample = q->upsample;
    a.n = q->n; a.n_pad = q->n_pad; a.k_pad = q->k_pad; a.out_scale = q->out_scale; a.flags = q->flags;
    a.zero_page = (const char*)q->zero_page;
    a.M = (long long)q->n_img * q->ho * q->wo;
    if (a.M <= 0 || a.M >= (1ll << 31)) return UAV_ESHAPE;
    a.a2_pix = 0;
    if (q->a2_images) {
        if (q->a2_images < 0 || q->c2 <= 0 || q->n_img != 2 * q->a2_images || q->upsample || q->stride != 1) return UAV_ESHAPE;
        a.a2_pix = q->a2_images * q->hi * q->wi;
    }
    a.a2_ctr = 0;
    if (q->a2_center_tap) {
        // the kernels that do not skip (128x128, ablation builds, channel-innermost order) still compute the same sum: the
        // off-centre weight entries of source 2 are zero by contract
        if (q->c2 <= 0 || q->kt != 1 || q->stride != 1 || q->upsample || q->pad_t != 0 || q->pad_h != q->kh / 2 || q->pad_w != q->kw / 2 ||
            !(q->kh & 1) || !(q->kw & 1) || small)
            return UAV_ESHAPE;
        a.a2_ctr = 1;
    }
    a.trace = nullptr;
    a.x1_bytes = a.x2_bytes = 0; a.dv_hw_mul = a.dv_wo_mul = a.dv_t_mul = 0; a.dv_hw_sh = a.dv_wo_sh = a.dv_t_sh = 32;
    a.lnp_raw = nullptr; a.lnp_stat = nullptr; a.lnc_stat = nullptr; a.lnc_colsum = nullptr; a.lnc_chunks = 0; a.lnc_n = 0; a.lnc_eps = 0.f;
    if (q->ln_raw_out || q->ln_stat_in) {
        if (!conv_ln_ok(q)) return UAV_ESHAPE;             // ask uav_conv_gemm_ln_ok() first
        if (q->ln_raw_out) { a.lnp_raw = (char*)q->ln_raw_out; a.lnp_stat = (float*)q->ln_stat_out; }
        if (q->ln_stat_in) {
            a.lnc_stat = q->ln_stat_in; a.lnc_colsum = q->ln_colsum; a.lnc_chunks = q->ln_chunks; a.lnc_n = q->ln_n; a.lnc_eps = q->ln_eps;
        }
    }
    a.omw = 0; a.omsy = 0; a.omsx = 0; a.omoff = 0;
    if (q->out_map_w > 0) {
        if (q->residual || (q->gn_partials && q->gn_chunk_cpi <= 0) || (q->flags & UAV_CONV_GEGLU) || q->out_map_sy < 0 ||
            q->out_map_sx <= 0 || q->out_map_off < 0)
            return UAV_ESHAPE;
        a.omw = q->out_map_w; a.omsy = q->out_map_sy; a.omsx = q->out_map_sx; a.omoff = q->out_map_off;
    } else if (q->out_map_w < 0) return UAV_ESHAPE;
    a.gn_ws = nullptr; a.gn_groups = 0; a.gn_cpg_log2 = 0; a.gn_chunks = 0; a.gn_cpi = 0; a.gn_cstride = 0; a.gn_coff = 0;
    if (q->gn_partials) {
        const int cl = conv_gn_cpg_log2(q);
        if (cl < 0) return UAV_ESHAPE;             // ask uav_conv_gemm_gn_chunk_rows() first
        a.gn_ws = (float*)q->gn_partials; a.gn_groups = q->gn_groups; a.gn_cpg_log2 = cl; a.gn_chunks = a.M / 64;
        if (q->gn_chunk_cpi > 0) {
            const long long k = a.M / 64;
            if ((k % q->gn_chunk_cpi) || q->gn_chunk_off < 0 || q->gn_chunk_off + q->gn_chunk_cpi > q->gn_chunk_stride ||
                (k / q->gn_chunk_cpi) * (long long)q->gn_chunk_stride > q->gn_chunks_total)
                return UAV_ESHAPE;
            a.gn_cpi = q->gn_chunk_cpi; a.gn_cstride = q->gn_chunk_stride; a.gn_coff = q->gn_chunk_off; a.gn_chunks = q->gn_chunks_total;
        }
    }
    // One-time setup.  The dynamic-LDS attribute of the 256x256 kernels and the CU count are PER DEVICE (std::call_once
    // per device index), so a second GPU, or a second host thread driving the library (bench --clips-per-step), never
    // launches before the attribute is in place.
    const ConvEnv& env = conv_env();
    a.korder = env.korder;
    a.tile_order = env.tile_order;
    if (!a.korder) a.a2_ctr = 0;                   // channel-innermost walk (A/B switch): multiply the zeros
    const long long mtiles = (a.M + BM - 1) / BM;
    const long long grid = mtiles * (q->n_pad / BN);
    if (grid >= (1ll << 31)) return UAV_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const long long mtiles256 = (a.M + LM - 1) / LM;
    const long long grid256 = mtiles256 * (q->n_pad / LN);
    const bool big = conv_uses_big_tile(q);
    const int gnm = a.gn_ws ? gn_mode_of(a.gn_cpg_log2) : 0;
    a.ntiles = (unsigned)grid256;
    if (big && a.M >= (1ll << 31)) return UAV_ESHAPE;
#ifdef UAV_DEV_KERNELS
    if (big && conv_uses_sk(q)) {
        const long long gsk = ((a.M + 127) / 128) * (q->n_pad / 256);        // 128 x 256 tiles
        if (gsk >= (1ll << 31)) return UAV_ESHAPE;
        a.ntiles = (unsigned)gsk;
        return conv_launch_dev(2, a, gsk, gnm, 0, 0, 0, env.sk == 2 ? 2 : 1, s);
    }
#endif
    if (big && conv_uses_w4(q)) {
        if (grid256 >= (1ll << 31)) return UAV_ESHAPE;
        const unsigned long long px = (unsigned long long)q->n_img * q->hi * q->wi;
        a.x1_bytes = (unsigned)(px * q->c1 * 2);
        a.x2_bytes = (unsigned)((q->a2_images ? px / 2 : px) * q->c2 * 2);
        conv_magic((unsigned)(q->ho * q->wo), &a.dv_hw_mul, &a.dv_hw_sh);
        conv_magic((unsigned)q->wo, &a.dv_wo_mul, &a.dv_wo_sh);
        conv_magic((unsigned)q->t_len, &a.dv_t_mul, &a.dv_t_sh);
#ifdef UAV_DEV_KERNELS
        static const bool w4_trace = getenv("UAV_CONV_W4_TRACE") != nullptr;
        if (w4_trace && gnm == 0 && !(q->flags & UAV_CONV_OUT_HILO)) return conv_launch_dev(3, a, grid256, 0, 0, 0, 0, 0, s);
#endif
        return conv_launch_wave4(a, grid256, gnm, (q->flags & UAV_CONV_OUT_HILO) != 0, s);
    }
    if (big) {
        if (grid256 >= (1ll << 31)) return UAV_ESHAPE;
#ifdef UAV_DEV_KERNELS
        if (a.lnp_raw || a.lnc_stat) return conv_launch_dev(1, a, grid256, 0, a.lnp_raw ? 1 : 2, 0, 0, 0, s);
        if (a.gn_ws && env.dmav != 6) return conv_launch_dev(1, a, grid256, gnm, 0, 0, 0, 0, s);
        if (!a.gn_ws && (env.dbg || env.dmav == 1 || env.persist || (q->flags & UAV_CONV_PERSISTENT)))
            return conv_launch_dev(env.dmav == 1 && !env.dbg ? 1 : 0, a, grid256, 0, 0, env.dbg, (env.persist || (q->flags & UAV_CONV_PERSISTENT)) ? 1 : 0, 0, s);
#endif
        return conv_launch_wave8(a, grid256, gnm, s);
    }
    return conv_launch_tile128(a, grid, small, s);
}
