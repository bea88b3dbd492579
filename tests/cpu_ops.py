"""TEST INFRASTRUCTURE ONLY — torch/CPU stand-ins for the entry points of `uav.ops`.

The product has no CPU path (`uav.ops` raises without the HIP library and a GPU, tests/test_cpu.py::test_no_cpu_fallback).
To exercise the HOST side of the engine — weight packing, the module orchestration in `models_video/*` (skip stack,
concatenation by two source pointers, forced upsample size, CFG-shared head, fused projections, GEGLU row order,
window / chunk loops of the pipeline) — on a box without a GPU, `install()` swaps the ops the models call for the
functions below, which follow the kernels' contracts (include/uav_hip.h): same packed-weight layout, same row
layouts, fp32 arithmetic, fp16 rounding of every stored tensor.  `restore()` puts the real entry points back.
Nothing under `upscale-a-video_amd/` imports this file.
"""
import math

import torch
import torch.nn.functional as F

HALF = torch.float16
_SAVED = {}


def _h(x):
    return x.to(HALF)


# ---------------------------------------------------------------------------------------------------------------------
def conv_gemm(a1, wt, *, n_img, t_len, hi, wi, stride=1, pad=None, upsample=False, a2=None, rowbias=None, rows_per_batch=0,
              residual=None, out_scale=1.0, out_f32=False, out=None, out_hw=None, persistent=False, act=None, gn_groups=None,
              out_map=None, a2_center=False, ln_produce=False, ln_consume=None, gn_shared=None, out_hilo=False):
    c1 = a1.shape[-1]
    c2 = 0 if a2 is None else a2.shape[-1]
    assert c1 + c2 == wt.cin_p, (c1, c2, wt.cin_p)
    if ln_consume is not None:               # LayerNorm folded in: rstd * (x16 . W'^T - mu * colsum) + bias, then GEGLU if asked
        src, colsum, eps = ln_consume
        assert a1 is src.raw and residual is None and rowbias is None and not out_f32
        s1 = src.stat[:, :, 0].sum(0); s2 = src.stat[:, :, 1].sum(0)
        mu = s1 / src.n; var = (s2 / src.n - mu * mu).clamp(min=0)
        rstd = torch.rsqrt(var + eps)
        acc = a1.float() @ wt.w[: wt.n, : src.n].float().t()
        y = rstd[:, None] * (acc - mu[:, None] * colsum[: wt.n].float()[None, :]) + wt.bias[: wt.n].float()
        if wt.geglu:
            yb = y.reshape(y.shape[0], wt.n // 64, 2, 32)
            y = (yb[:, :, 0] * F.gelu(yb[:, :, 1])).reshape(y.shape[0], wt.n // 2)
        return _h(y * out_scale)
    assert a1.dtype == HALF and a1.numel() == n_img * hi * wi * c1
    if pad is None:
        pad = (wt.kt // 2, wt.kh // 2, wt.kw // 2)
    pt, ph, pw = pad
    if a2 is not None and a2.shape[0] * 2 == a1.shape[0]:                             # skip tensor read batch-broadcast
        a2 = torch.cat([a2, a2])
    x = a1.float() if a2 is None else torch.cat([a1.float(), a2.float()], dim=-1)
    nb = n_img // t_len
    x = x.reshape(nb, t_len, hi, wi, wt.cin_p).permute(0, 4, 1, 2, 3)                 # (B, C, T, H, W)
    if upsample:
        x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
        ho, wo = 2 * hi, 2 * wi
    else:
        ho = (hi + 2 * ph - wt.kh) // stride + 1
        wo = (wi + 2 * pw - wt.kw) // stride + 1
    if out_hw is not None:                                                            # taps past the edge read zeros
        ho, wo = out_hw
    hin, win = x.shape[-2:]
    need_h = (ho - 1) * stride + wt.kh - 2 * ph
    need_w = (wo - 1) * stride + wt.kw - 2 * pw
    if need_h > hin or need_w > win:
        x = F.pad(x, (0, max(0, need_w - win), 0, max(0, need_h - hin)))
    ntaps = wt.kt * wt.kh * wt.kw
    w = wt.w[: wt.n, : ntaps * wt.cin_p].float().reshape(wt.n, wt.kt, wt.kh, wt.kw, wt.cin_p).permute(0, 4, 1, 2, 3)
    y = F.conv3d(x, w, None, stride=(1, stride, stride), padding=(pt, ph, pw))[..., :ho, :wo]
    assert y.shape[2:] == (t_len, ho, wo), (y.shape, t_len, ho, wo)
    y = y.permute(0, 2, 3, 4, 1).reshape(n_img * ho * wo, wt.n)
    m = y.shape[0]
    if wt.bias is not None:
        y = y + wt.bias[: wt.n].float()
    if rowbias is not None:
        idx = torch.arange(m) // rows_per_batch
        y = y + rowbias.float()[idx, : wt.n]
    if wt.geglu:
        yb = y.reshape(m, wt.n // 64, 2, 32)                                          # [32 value | 32 gate] blocks
        y = (yb[:, :, 0] * F.gelu(yb[:, :, 1])).reshape(m, wt.n // 2)
    if act == "gelu":
        y = F.gelu(y)
    elif act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    if residual is not None:
        assert residual.shape[0] == m
        y = y + residual.float()[:, : y.shape[1]]
    y = y * out_scale
    if out_hilo:                             # the fp32 result as its fp16 operand pair (UAV_CONV_OUT_HILO == cast_hilo of it)
        assert out_f32 and out is None and out_map is None
        return cast_hilo(y)
    res = y if out_f32 else _h(y)
    if ln_produce and out_f32 and res.shape[1] % 128 == 0 and out is None:
        from uav import ops as _ops
        nc = res.shape[1] // 128
        yc = res.reshape(res.shape[0], nc, 128)
        op = _ops.LnOperand(_h(res), torch.stack([yc.sum(-1).t(), (yc * yc).sum(-1).t()], dim=-1).contiguous(), res.shape[1])
        op.version = res._version
        res._uav_ln = op
    if out_map is not None:                  # strided output rows (sub-pixel phases of the upsampling convs)
        w_, sy, sx, off = out_map
        mm = torch.arange(res.shape[0])
        out[(mm // w_) * sy + (mm % w_) * sx + off, : res.shape[1]] = res
        return out
    if out is not None:                      # the kernel writes n_out columns of a possibly wider row (out_stride)
        out[:, : res.shape[1]].copy_(res)
        return out
    return res


def _factor_rows(m):
    return 1, m


def linear(x, wt, *, residual=None, out_scale=1.0, rowbias=None, rows_per_batch=0, out_f32=False, act=None, gn_groups=None,
           ln_produce=False, ln_consume=None, out_hilo=False):
    return conv_gemm(x, wt, n_img=1, t_len=1, hi=x.shape[0], wi=1, residual=residual, out_scale=out_scale, rowbias=rowbias,
                     rows_per_batch=rows_per_batch, out_f32=out_f32, act=act, ln_produce=ln_produce, ln_consume=ln_consume,
                     out_hilo=out_hilo)


def ln_fold_ok(m, k, wt):
    """Stand-in for the host-side launch query: the CPU tests fold wherever the layout allows (the GPU library additionally
    asks for the 256x256 kernel, i.e. large launches)."""
    return k % 128 == 0 and wt.n % 128 == 0


# ---------------------------------------------------------------------------------------------------------------------
def groupnorm(x1, gamma, beta, *, n_inst, rows_per_inst, groups, eps, silu, x2=None, c_real=None, want_raw=False):
    if x2 is not None and x2.shape[0] * 2 == x1.shape[0]:                             # skip tensor read batch-broadcast
        x2 = torch.cat([x2, x2])
    x = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=-1)
    if want_raw:
        hi = _h(x)
        raw = torch.cat([hi, _h(x - hi.float())], dim=-1) if want_raw == "hilo" else hi
        return groupnorm(x1, gamma, beta, n_inst=n_inst, rows_per_inst=rows_per_inst, groups=groups, eps=eps, silu=silu,
                         x2=x2, c_real=c_real), raw
    c = x.shape[-1]
    c_real = c if c_real is None else c_real
    xr = x[:, :c_real].reshape(n_inst, rows_per_inst, groups, c_real // groups).double()
    mean = xr.mean(dim=(1, 3), keepdim=True)
    var = xr.var(dim=(1, 3), unbiased=False, keepdim=True)
    y = ((xr - mean) / torch.sqrt(var + eps)).float().reshape(n_inst * rows_per_inst, c_real)
    y = y * gamma.float()[:c_real] + beta.float()[:c_real]
    if silu:
        y = F.silu(y)
    if c_real < c:
        y = F.pad(y, (0, c - c_real))
    return _h(y)


def groupnorm_scale_shift(x1, gamma, beta, *, n_inst, rows_per_inst, groups, eps, x2=None, c_real=None):
    """Per-instance rows [n_inst][c]: y = x * scale + shift (single fp32 source only: what the fused transformer launch asks for)."""
    assert x2 is None and c_real is None
    c = x1.shape[-1]
    xr = x1.float().reshape(n_inst, rows_per_inst, groups, c // groups).double()
    mean = xr.mean(dim=(1, 3)); var = xr.var(dim=(1, 3), unbiased=False)            # [n_inst][groups]
    rstd = 1.0 / torch.sqrt(var + eps)
    scale = (rstd.repeat_interleave(c // groups, 1) * gamma.double()).float()
    shift = (beta.double() - (mean * rstd).repeat_interleave(c // groups, 1) * gamma.double()).float()
    return scale.contiguous(), shift.contiguous()


def groupnorm_apply(x1, scale, shift, *, n_inst, rows_per_inst, silu, x2=None, want_raw=False):
    assert x2 is None and not want_raw
    c = x1.shape[-1]
    y = x1.float().reshape(n_inst, rows_per_inst, c) * scale[:, None, :] + shift[:, None, :]
    y = y.reshape(-1, c)
    return _h(F.silu(y) if silu else y)


def layernorm(x, gamma, beta, eps=1e-5):
    return _h(F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps))


# ---------------------------------------------------------------------------------------------------------------------
def attention(q, k, v, *, bq, lq, lk, heads, head_dim, q_per_kv=1, scale=None, q_stride=None, k_stride=None, v_stride=None,
              causal=False):
    c = heads * head_dim
    scale = head_dim ** -0.5 if scale is None else scale
    qh = q.float().reshape(bq, lq, heads, head_dim).permute(0, 2, 1, 3)
    kh = k.float().reshape(bq // q_per_kv, lk, heads, head_dim).repeat_interleave(q_per_kv, 0).permute(0, 2, 1, 3)
    vh = v.float().reshape(bq // q_per_kv, lk, heads, head_dim).repeat_interleave(q_per_kv, 0).permute(0, 2, 1, 3)
    sc = qh @ kh.transpose(-1, -2) * scale
    if causal:
        sc = sc + torch.full((lq, lk), float("-inf")).triu(1)
    p = torch.softmax(sc, dim=-1)
    return _h((p @ vh).permute(0, 2, 1, 3).reshape(bq * lq, c))


# ---------------------------------------------------------------------------------------------------------------------
# Fused text cross-attention sub-layer (csrc/xattn_fused.hip).  The stand-ins DECODE the MFMA-fragment streams (uav.ops.pack_xattn_weight
# and the layout uav_xattn_pack_kv writes) back to plain matrices, so the host-side packing is checked on CPU too.
def _unpack_xattn_weight(packed, kind):
    if kind == "q":        # [h][ks][mt][hi][l32][e2][e1] -> rows (h, mt, l32) x k (ks, e2, hi, e1)
        return packed.float().reshape(8, 32, 2, 2, 32, 2, 4).permute(0, 2, 4, 1, 5, 3, 6).reshape(512, 512)
    # [h][npair][ks][par][hi][l32][e2][e1] -> rows (npair, par, l32) x k (h, ks, e2, hi, e1)
    return packed.float().reshape(8, 8, 4, 2, 2, 32, 2, 4).permute(1, 3, 5, 0, 2, 6, 4, 7).reshape(512, 512)


def xattn_pack_kv(k, v, *, n_batch, lk, k_stride=None, v_stride=None):
    """[n_batch][8 heads][32 fragments][64 lanes][8 halves]: 0..11 K (key tile f % 3, k-step f / 3), 12..23 V^T (k-step g >> 1, tile g & 1)."""
    out = torch.zeros(n_batch, 8, 32, 2, 32, 2, 4, dtype=HALF)                       # (b, h, f, hi, l32, e2, e1)
    kk = torch.zeros(n_batch, 96, 8, 64, dtype=HALF); vv = torch.zeros(n_batch, 96, 8, 64, dtype=HALF)
    kk[:, :lk] = k.reshape(n_batch, lk, 8, 64); vv[:, :lk] = v.reshape(n_batch, lk, 8, 64)
    # K fragment f = kt + 3 ks: rows key = 32 kt + l32, k = channel 16 ks + 8 e2 + 4 hi + e1
    kf = kk.reshape(n_batch, 3, 32, 8, 4, 2, 2, 4)                                    # (b, kt, l32, h, ks, e2, hi, e1)
    out[:, :, :12] = kf.permute(0, 3, 4, 1, 6, 2, 5, 7).reshape(n_batch, 8, 12, 2, 32, 2, 4)      # (b, h, ks, kt, hi, l32, e2, e1): f = 3 ks + kt
    # V^T fragment g = 2 ks + mt: rows channel 32 mt + l32, k = key 16 ks + 8 e2 + 4 hi + e1
    vf = vv.reshape(n_batch, 6, 2, 2, 4, 8, 2, 32)                                    # (b, ks, e2, hi, e1, h, mt, l32)
    out[:, :, 12:24] = vf.permute(0, 5, 1, 6, 3, 7, 2, 4).reshape(n_batch, 8, 12, 2, 32, 2, 4)    # (b, h, ks, mt, hi, l32, e2, e1)
    return out.reshape(n_batch, 8, 32 * 64 * 8)


def _unpack_xattn_kv(kvp, lk):
    n_batch = kvp.shape[0]
    f = kvp.reshape(n_batch, 8, 32, 2, 32, 2, 4)
    kf = f[:, :, :12].reshape(n_batch, 8, 4, 3, 2, 32, 2, 4)                          # (b, h, ks, kt, hi, l32, e2, e1)
    k = kf.permute(0, 3, 5, 1, 2, 6, 4, 7).reshape(n_batch, 96, 8, 64)                # key (kt, l32), head, channel (ks, e2, hi, e1)
    vf = f[:, :, 12:24].reshape(n_batch, 8, 6, 2, 2, 32, 2, 4)                        # (b, h, ks, mt, hi, l32, e2, e1)
    v = vf.permute(0, 2, 6, 4, 7, 1, 3, 5).reshape(n_batch, 96, 8, 64)                # key (ks, e2, hi, e1), head, channel (mt, l32)
    return k[:, :lk].float(), v[:, :lk].float()


def xattn_sublayers(x, subs, *, rows_per_kv, lk, scale, out=None):
    y = x
    for sub in subs:
        y = xattn_sublayer(y, *sub, rows_per_kv=rows_per_kv, lk=lk, scale=scale)
    if out is not None:
        out.copy_(y)
        return out
    return y


def xattn_sublayer(x, gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias, *, rows_per_kv, lk, scale, out=None):
    wq = _unpack_xattn_weight(wq_packed, "q"); wo = _unpack_xattn_weight(wo_packed, "out")
    k, v = _unpack_xattn_kv(kv_packed, lk)                                            # (B, lk, 8, 64)
    m = x.shape[0]
    nb = m // rows_per_kv
    n = _h(F.layer_norm(x.float(), (512,), gamma.float(), beta.float(), eps)).float()
    q = _h(n @ wq.t()).float().reshape(nb, rows_per_kv, 8, 64).permute(0, 2, 1, 3)
    sc = q @ k.permute(0, 2, 3, 1) * scale                                            # (B, 8, rows, lk)
    sc = sc - sc.amax(-1, keepdim=True)
    e = torch.exp(sc)
    o = (_h(e).float() @ v.permute(0, 2, 1, 3)) / e.sum(-1, keepdim=True)             # P rounded to fp16 unnormalised, like the kernel
    o = _h(o.permute(0, 2, 1, 3).reshape(m, 512)).float()
    y = x.float() + out_bias.float() + o @ wo.t()
    if out is not None:
        out.copy_(y)
        return out
    return y


def temporal_attention(qkv, *, n_batch, t_len, hw, c, heads, scale, rope_cos, rope_sin, rot_dim, bias):
    d = c // heads
    x = qkv.float().reshape(n_batch, t_len, hw, 3, heads, d)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]                             # (B, T, P, H, d)
    q = q * scale

    def rope(t):
        if rot_dim == 0:
            return t
        cos = rope_cos.float()[:t_len].reshape(1, t_len, 1, 1, rot_dim // 2)
        sin = rope_sin.float()[:t_len].reshape(1, t_len, 1, 1, rot_dim // 2)
        a, b = t[..., 0:rot_dim:2], t[..., 1:rot_dim:2]
        r = torch.stack([a * cos - b * sin, b * cos + a * sin], dim=-1).reshape(t.shape[:-1] + (rot_dim,))
        return torch.cat([r, t[..., rot_dim:]], dim=-1)
    q, k = _h(rope(q)).float(), _h(rope(k)).float()                                   # rotary output is fp16 (as the kernel)
    q, k, v = (t.permute(0, 2, 3, 1, 4) for t in (q, k, v))                           # (B, P, H, T, d)
    s = q @ k.transpose(-1, -2) + bias.float().reshape(1, 1, heads, t_len, t_len)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).permute(0, 3, 1, 2, 4).reshape(n_batch * t_len * hw, c)               # rows (b, t, p)
    return _h(o)


def _attach_next_ln(y, next_ln):
    if next_ln is None:
        return y
    from uav import ops as _ops
    gamma, beta, eps = next_ln
    nl = _ops.NextLn(_h(F.layer_norm(y.float(), (y.shape[-1],), gamma.float(), beta.float(), eps)), gamma, beta, eps)
    nl.version = y._version
    y._uav_next_ln = nl
    return y


def tattn_sublayer(x, gamma, beta, eps, wq_packed, wk_packed, wv_packed, wo_packed, out_bias, rel_bias, rope_cos, rope_sin, *,
                   n_batch, t_len, hw, rot_dim, scale, out=None, next_ln=None):
    """Fused temporal sub-layer (csrc/xattn_fused.hip tattn_sublayer_kernel): the chain LayerNorm -> q | k | v -> temporal attention ->
    to_out + residual on the decoded fragment streams."""
    wq, wk, wv = (_unpack_xattn_weight(w, "q") for w in (wq_packed, wk_packed, wv_packed))
    wo = _unpack_xattn_weight(wo_packed, "out")
    n = _h(F.layer_norm(x.float(), (512,), gamma.float(), beta.float(), eps)).float()
    qkv = _h(torch.cat([n @ wq.t(), n @ wk.t(), n @ wv.t()], dim=-1))
    o = temporal_attention(qkv, n_batch=n_batch, t_len=t_len, hw=hw, c=512, heads=8, scale=scale, rope_cos=rope_cos, rope_sin=rope_sin,
                           rot_dim=rot_dim, bias=rel_bias)
    y = x.float() + out_bias.float() + o.float() @ wo.t()
    if out is not None:
        out.copy_(y)
        return _attach_next_ln(out, next_ln)
    return _attach_next_ln(y, next_ln)


def block_attn_sublayers(x, cross, temporal, *, n_batch, t_len, hw, lk, cross_scale, temporal_scale, out=None, next_ln=None):
    y = xattn_sublayers(x, cross, rows_per_kv=t_len * hw, lk=lk, scale=cross_scale)
    y = tattn_sublayer(y, *temporal[:11], n_batch=n_batch, t_len=t_len, hw=hw, rot_dim=temporal[11], scale=temporal_scale)
    if out is not None:
        out.copy_(y)
        return _attach_next_ln(out, next_ln)
    return _attach_next_ln(y, next_ln)


# ---------------------------------------------------------------------------------------------------------------------
def linear_small(x, w, b, *, pre_silu=False, post_silu=False):
    x = F.silu(x.float()) if pre_silu else x.float()
    y = x @ w.float().t() + (0 if b is None else b.float())
    return F.silu(y) if post_silu else y


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0):
    half = dim // 2
    expo = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    arg = t.float()[:, None] * torch.exp(expo)[None, :]
    sn, cs = torch.sin(arg), torch.cos(arg)
    return torch.cat([cs, sn], dim=-1) if flip_sin_to_cos else torch.cat([sn, cs], dim=-1)


def pack_nhwc(src1, src2=None, c_pad=8, scale=1.0):
    x = src1.float() if src2 is None else torch.cat([src1.float(), src2.float()], dim=1)
    b, c, t, h, w = x.shape
    rows = (x * scale).permute(0, 2, 3, 4, 1).reshape(b * t * h * w, c)
    return _h(F.pad(rows, (0, c_pad - c)))


def unpack_ncthw(src, *, c, n_batch, t_len, h, w, out_dtype=HALF, clamp=None):
    x = src.float()[:, :c].reshape(n_batch, t_len, h, w, c).permute(0, 4, 1, 2, 3)
    if clamp is not None:
        x = x.clamp(clamp[0], clamp[1])
    return x.to(out_dtype).contiguous()


def _like(y, x):
    """scheduler kernels: fp16 in -> fp16 out (rounded), fp32 in -> fp32 out (uav_*_f32 entry points)"""
    return y.float() if x.dtype == torch.float32 else _h(y)


def axpby(x, z, a, b):
    return _like(a * x.float() + b * z.float(), x)


def cfg_ddim_v0(eu, ec, sample, *, guidance, coef_sample, coef_eps, clip=False, clip_range=1.0):
    g = eu.float() if ec is None else eu.float() + guidance * (ec.float() - eu.float())
    g = _like(g, sample)
    x0 = coef_sample * sample.float() + coef_eps * g.float()
    if clip:
        x0 = x0.clamp(-clip_range, clip_range)
    return g, _like(x0, sample)


def ddim_vt(x0, guided, sample, *, coef_x0, coef_dir, eps_from_model, eps_from_sample, eps_from_x0=0.0, clip=False,
            clip_range=1.0):
    a = x0.float().clamp(-clip_range, clip_range) if clip else x0.float()
    eps = eps_from_model * guided.float() + eps_from_sample * sample.float() + eps_from_x0 * a
    return _like(coef_x0 * a + coef_dir * eps, sample)


def _warp(x, flow, mode):
    """flow_warp of the reference (propagation_module.py:104-135): x (1,C,h,w), flow (1,2,h,w) in pixels."""
    _, _, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    vx, vy = gx + flow[0, 0], gy + flow[0, 1]
    grid = torch.stack((2.0 * vx / max(w - 1, 1) - 1.0, 2.0 * vy / max(h - 1, 1) - 1.0), dim=-1)[None]
    return F.grid_sample(x, grid, mode=mode, padding_mode="zeros", align_corners=True)


def propagate_step(feat_prev, feat_cur, flow_prop, flow_check, out, *, c, h, w, feat_chan_stride, flow_chan_stride, nearest,
                   coord_f16, fuse_scale, alpha1, alpha2):
    """One recurrence step (uav_propagate_step_f16) with fp32 coordinates: consistency mask from the bilinearly warped
    check flow, nearest / bilinear warp of the propagated frame, fuse, mask blend; written into the `out` frame view."""
    fp, fc = flow_prop.float()[None], flow_check.float()[None]
    bw = _warp(fc, fp, "bilinear")
    lsq = lambda t: (t * t).sum(dim=1, keepdim=True)
    mask = (lsq(fp + bw) < alpha1 * (lsq(fp) + lsq(bw)) + alpha2).float()
    cur = feat_cur.float()[None]
    warped = _warp(feat_prev.float()[None], fp, "nearest" if nearest else "bilinear")
    fused = fuse_scale * warped + (1.0 - fuse_scale) * cur
    res = mask * fused + (1.0 - mask) * cur
    out.copy_((_h(res) if out.dtype == HALF else res)[0])
    return out


def resize_area_f32(x, ho, wo, mul=1.0):
    lead = x.shape[:-2]
    return (F.interpolate(x.reshape((-1, 1) + tuple(x.shape[-2:])).float(), (ho, wo), mode="area") * mul).reshape(tuple(lead) + (ho, wo))


def _unpack_ff_weights(w_packed):
    """Inverse of uav.ops.pack_ff_weights: -> (w_up [4096][512], w_down [512][2048])."""
    f = w_packed.float().reshape(64, 3 * 32 * 512)
    # [c][ks][vg][hi][l32][e2][e1] -> rows (vg, c, l32) x k (ks, e2, hi, e1)
    wu = f[:, :2 * 32 * 512].reshape(64, 32, 2, 2, 32, 2, 4).permute(2, 0, 4, 1, 5, 3, 6).reshape(4096, 512)
    # [c][npair][ks][par][hi][l32][e2][e1] -> rows (npair, par, l32) x k (c, ks, e2, hi, e1)
    wd = f[:, 2 * 32 * 512:].reshape(64, 8, 2, 2, 2, 32, 2, 4).permute(1, 3, 5, 0, 2, 6, 4, 7).reshape(512, 2048)
    return wu, wd


def ff_sublayer(x, gamma, beta, eps, w_packed, up_bias, down_bias, *, out_f32=True, out_hilo=False):
    wu, wd = _unpack_ff_weights(w_packed)
    n = _h(F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)).float()
    u = n @ wu.t() + up_bias.float()
    hdn = _h(u[:, :2048] * F.gelu(u[:, 2048:])).float()
    y = x.float() + down_bias.float() + hdn @ wd.t()
    if out_f32 and out_hilo:
        return y, cast_hilo(y)
    return y if out_f32 else cast_hilo(y)


def block_sublayers(x, cross, temporal, ff, *, n_batch, t_len, hw, lk, cross_scale, temporal_scale, out_f32=True, out_hilo=False, proj_in=None):
    if proj_in is not None:
        sc, sh, wp, bi = proj_in
        n = _h(groupnorm_apply(x, sc, sh, n_inst=n_batch * t_len, rows_per_inst=hw, silu=False)).float()
        x = n @ _unpack_xattn_weight(wp, "out").t() + bi.float()
    y = block_attn_sublayers(x, cross, temporal, n_batch=n_batch, t_len=t_len, hw=hw, lk=lk, cross_scale=cross_scale, temporal_scale=temporal_scale)
    return ff_sublayer(y, *ff, out_f32=out_f32, out_hilo=out_hilo)


def cast_f16(x):
    return x if x.dtype == HALF else _h(x)


def cast_hilo(x):
    hi = _h(x)
    return torch.cat([hi, _h(x.float() - hi.float())], dim=-1)


def sft_fuse(dec, scale, shift, w, out_f32=False):
    assert dec.dtype == scale.dtype == shift.dtype
    d = dec.float()
    y = d + w * (d * scale.float() + shift.float())
    return y if out_f32 else _h(y)


_OPS = ("ln_fold_ok", "resize_area_f32", "cast_f16", "cast_hilo", "sft_fuse", "propagate_step", "conv_gemm", "linear", "groupnorm", "groupnorm_scale_shift", "groupnorm_apply", "layernorm", "attention", "xattn_pack_kv", "xattn_sublayer", "xattn_sublayers", "tattn_sublayer", "block_attn_sublayers", "block_sublayers", "ff_sublayer", "temporal_attention", "linear_small",
        "timestep_embedding", "pack_nhwc", "unpack_ncthw", "axpby", "cfg_ddim_v0", "ddim_vt")


def install():
    """Swap the uav.ops entry points (and the engine's device check) for the CPU stand-ins.  Not re-entrant."""
    from uav import engine, ops
    assert not _SAVED, "cpu_ops.install() is already active"
    for name in _OPS:
        _SAVED[name] = getattr(ops, name)
        setattr(ops, name, globals()[name])
    _SAVED["_dev"] = engine._dev
    engine._dev = lambda p: p.device


def restore():
    from uav import engine, ops
    for name in _OPS:
        if name in _SAVED:
            setattr(ops, name, _SAVED.pop(name))
    if "_dev" in _SAVED:
        engine._dev = _SAVED.pop("_dev")
