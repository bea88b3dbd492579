// The temporal attention sub-layer and the three attention sub-layers of a block as single launches (kernel: tattn_kernel.h, design notes:
// xattn_fused.hip); the whole-block instance with the feed-forward is tattn_block_fused.hip.
#include "tattn_kernel.h"

int uav_tattn_run_attn(const void* tattn_args, int nx, dim3 grid, hipStream_t stream) {
    const TattnArgs& a = *(const TattnArgs*)tattn_args;
    if (nx) {
        static UavDynLds lds2;
        if (int rc = uav_set_dyn_lds(lds2, (const void*)tattn_sublayer_kernel<2, 0>, TSMEM)) return rc;
        hipLaunchKernelGGL((tattn_sublayer_kernel<2, 0>), grid, dim3(256), TSMEM, stream, a);
    } else {
        static UavDynLds lds0;
        if (int rc = uav_set_dyn_lds(lds0, (const void*)tattn_sublayer_kernel<0, 0>, TSMEM)) return rc;
        hipLaunchKernelGGL((tattn_sublayer_kernel<0, 0>), grid, dim3(256), TSMEM, stream, a);
    }
    return uav_launch_status();
}

namespace {
int tattn_launch(const float* x, float* out, const uav_xattn_params* xs, int32_t n_xs, int32_t lk, float xscale, const uav_tattn_params* q,
                 int32_t n_batch, int32_t t_len, int64_t hw, int32_t channels, int32_t heads, float scale, void* stream,
                 const uav_ff_params* ff = nullptr, void* out_hilo = nullptr, const uav_projin_params* pi = nullptr) {
    if (!x || (!out && !(ff && out_hilo)) || !q || !q->ln_gamma || !q->ln_beta || !q->wq_packed || !q->wk_packed || !q->wv_packed || !q->wo_packed || !q->out_bias ||
        !q->rel_bias || !q->rope_cos || !q->rope_sin)
        return UAV_EINVAL;
    if (channels != XC || heads != XHEADS || t_len != TT || q->rot_dim != 32) return UAV_ESHAPE;
    if (n_batch <= 0 || hw <= 0 || (hw % 16) || (long long)n_batch * (hw / 16) >= (1ll << 31)) return UAV_ESHAPE;
    if (n_xs != 0 && n_xs != 2) return UAV_ESHAPE;
    if (n_xs && (!xs || lk <= 0 || lk > 96)) return UAV_ESHAPE;
    if (((size_t)x | (size_t)out) & 15) return UAV_EALIGN;
    TattnArgs a{x, out, q->ln_gamma, q->ln_beta, q->out_bias, q->ln_eps, (const char*)q->wq_packed, (const char*)q->wk_packed,
                (const char*)q->wv_packed, (const char*)q->wo_packed, q->rel_bias, q->rope_cos, q->rope_sin, n_batch, (long long)hw, scale,
                {}, lk, xscale * 1.44269504088896341f, (half_t*)q->next_ln_out, q->next_ln_gamma, q->next_ln_beta, q->next_ln_eps,
                nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (ff) {
        if (n_xs != 2 || !ff->ln_gamma || !ff->ln_beta || !ff->w_packed || !ff->up_bias || !ff->down_bias || q->next_ln_out) return UAV_EINVAL;
        if (((size_t)out_hilo | (size_t)ff->up_bias | (size_t)ff->down_bias | (size_t)ff->w_packed) & 15) return UAV_EALIGN;
        a.ff_gamma = ff->ln_gamma; a.ff_beta = ff->ln_beta; a.ff_down_bias = ff->down_bias; a.ff_up_bias = ff->up_bias;
        a.ff_w = (const char*)ff->w_packed; a.ff_eps = ff->ln_eps; a.out_hilo = (half_t*)out_hilo;
    }
    if (q->next_ln_out && (!q->next_ln_gamma || !q->next_ln_beta || ((size_t)q->next_ln_out & 15))) return UAV_EINVAL;
    for (int i = 0; i < n_xs; ++i) {
        const uav_xattn_params& c = xs[i];
        if (!c.ln_gamma || !c.ln_beta || !c.wq_packed || !c.kv_packed || !c.wo_packed || !c.out_bias) return UAV_EINVAL;
        a.xs[i] = XattnSub{c.ln_gamma, c.ln_beta, c.out_bias, (const char*)c.wq_packed, (const char*)c.kv_packed, (const char*)c.wo_packed, c.ln_eps};
    }
    const dim3 grid((unsigned)(n_batch * (hw / 16)));
    if (pi) {
        if (!ff || !pi->gn_scale || !pi->gn_shift || !pi->w_packed || !pi->bias) return UAV_EINVAL;
        if (((size_t)pi->gn_scale | (size_t)pi->gn_shift | (size_t)pi->w_packed | (size_t)pi->bias) & 15) return UAV_EALIGN;
        a.gn_scale = pi->gn_scale; a.gn_shift = pi->gn_shift; a.w_in = (const char*)pi->w_packed; a.b_in = pi->bias;
        return uav_tattn_run_block_pi(&a, grid, (hipStream_t)stream);
    }
    if (ff) return uav_tattn_run_block_ff(&a, grid, (hipStream_t)stream);
    return uav_tattn_run_attn(&a, n_xs, grid, (hipStream_t)stream);
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

extern "C" int uav_block_sublayers_f32(const float* x, const uav_projin_params* proj_in, float* out, void* out_hilo, const uav_xattn_params* cross,
                                       int32_t lk, float cross_scale, const uav_tattn_params* temporal, const uav_ff_params* ff, int32_t n_batch,
                                       int32_t t_len, int64_t hw, int32_t channels, int32_t heads, int32_t inner, float temporal_scale,
                                       void* stream) {
    if (!ff) return UAV_EINVAL;
    if (inner != FINNER) return UAV_ESHAPE;
    return tattn_launch(x, out, cross, 2, lk, cross_scale, temporal, n_batch, t_len, hw, channels, heads, temporal_scale, stream, ff, out_hilo,
                        proj_in);
}
