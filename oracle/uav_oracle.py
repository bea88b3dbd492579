"""TEST INFRASTRUCTURE ONLY — the parity oracle.  Never imported by the product path
(`upscale-a-video_amd/`); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may use it, and only as the checker.

CPU fp32 restatement (plain PyTorch functional ops, channels-first like the reference) of the hot
path of sczhou/Upscale-A-Video: UNetVideoModel.forward, AutoencoderKLVideo.decode, the DDIM
scheduler split (step_v0 / step_vt), flow-guided Propagation and the VideoUpscalePipeline loop.
Every function cites the reference lines (relative to /root/reference/) it restates.  It works on
a state dict with the REFERENCE's key names, so the same synthetic weights feed the reference
module, this oracle and the HIP engine.

Pinning: the reference ships no tests / golden vectors.  This restatement is
pinned against the reference's OWN modules, imported unmodified under oracle/ref_stubs.py in the
build container, by oracle/make_golden.py (max-abs agreement recorded in
tests/golden/PINNING.json) and the resulting vectors are committed under tests/golden/.
Third-party arithmetic absent from /root/reference (diffusers 0.16.0 Timesteps /
TimestepEmbedding / AttentionBlock / GEGLU, rotary-embedding-torch 0.2.3) is restated from the
reference's vendored spec copy (models_video/diffusers_attention.py) and the call sites; see
ref_stubs.py for what that implies.
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F


class P:
    """State-dict view with a key prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def get(self, k):
        return self.sd.get(self.prefix + k)

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, name):
        return P(self.sd, self.prefix + name + ".")


# ---------------------------------------------------------------------------------------------
# building blocks
def conv2d_frames(x, p, stride=1, padding=1):
    """InflatedConv3d = nn.Conv2d applied per frame (models_video/resnet.py:94-101)."""
    b, c, t, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), p["weight"], p.get("bias"), stride, padding)
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def group_norm(x, p, groups, eps):
    """nn.GroupNorm; on a 5-D tensor the statistics span (C/G, T, H, W)."""
    return F.group_norm(x, groups, p["weight"], p["bias"], eps)


def temb_proj(p, temb):
    """time_emb_proj(nonlinearity(temb))[:, :, None, None, None] (resnet.py:272-273)."""
    return F.linear(F.silu(temb), p["time_emb_proj.weight"], p["time_emb_proj.bias"])[:, :, None, None, None]


def resnet_block3d(x, p, temb=None, groups=32, groups_out=None, eps=1e-6, output_scale_factor=1.0):
    """ResnetBlock3D.forward (resnet.py:264-294)."""
    groups_out = groups_out or groups
    h = F.silu(group_norm(x, p.sub("norm1"), groups, eps))
    h = conv2d_frames(h, p.sub("conv1"))
    if temb is not None and p.has("time_emb_proj.weight"):
        h = h + temb_proj(p, temb)
    h = F.silu(group_norm(h, p.sub("norm2"), groups_out, eps))
    h = conv2d_frames(h, p.sub("conv2"))
    if p.has("conv_shortcut.weight"):
        x = conv2d_frames(x, p.sub("conv_shortcut"), padding=0)
    return (x + h) / output_scale_factor


def resnet_block3d_cnn(x, p, temb=None, kt=3, groups=32, eps=1e-6):
    """ResnetBlock3DCNN.forward (resnet.py:363-393): Conv3d (kt,1,1) then (3,1,1)."""
    h = F.silu(group_norm(x, p.sub("norm1"), groups, eps))
    h = F.conv3d(h, p["conv1.weight"], p["conv1.bias"], padding=(kt // 2, 0, 0))
    if temb is not None and p.has("time_emb_proj.weight"):
        h = h + temb_proj(p, temb)
    h = F.silu(group_norm(h, p.sub("norm2"), groups, eps))
    h = F.conv3d(h, p["conv2.weight"], p["conv2.bias"], padding=(1, 0, 0))
    if p.has("conv_shortcut.weight"):
        x = F.conv3d(x, p["conv_shortcut.weight"], p["conv_shortcut.bias"])
    return x + h


def resnet_block3d_plus(x, p, groups=32, groups_out=None, eps=1e-6):
    """ResnetBlock3D_plus.forward (resnet.py:464-500): ResnetBlock3D + GN/SiLU/Conv3d 3x3x3 branch."""
    groups_out = groups_out or 32
    out = resnet_block3d(x, p, None, groups, groups_out, eps)
    h = F.silu(group_norm(out, p.sub("norm_3d"), groups_out, eps))
    h = F.conv3d(h, p["conv_3d.weight"], p["conv_3d.bias"], padding=1)
    return out + h


def _heads(x, heads):
    b, l, c = x.shape
    return x.reshape(b, l, heads, c // heads).permute(0, 2, 1, 3)


def cross_attention(x, ctx, p, heads):
    """CrossAttention.forward/_attention (attention.py:148-238), no mask, softmax(QK^T d^-0.5)V."""
    ctx = x if ctx is None else ctx
    q = _heads(F.linear(x, p["to_q.weight"]), heads)
    k = _heads(F.linear(ctx, p["to_k.weight"]), heads)
    v = _heads(F.linear(ctx, p["to_v.weight"]), heads)
    d = q.shape[-1]
    a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[1], -1)
    return F.linear(o, p["to_out.0.weight"], p["to_out.0.bias"])


def relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """RelativePositionBias._relative_position_bucket (attention.py:747-765)."""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def time_rel_pos_bias(weight, n, max_distance=32):
    """RelativePositionBias.forward (attention.py:767-772) with heads from the embedding table;
    TemporalAttention builds it with max_distance=32 (attention.py:641)."""
    pos = torch.arange(n)
    rel = pos[None, :] - pos[:, None]
    bucket = relative_position_bucket(rel, 32, max_distance)
    return weight[bucket].permute(2, 0, 1)            # (heads, n, n)


def rotary(x, freqs):
    """RotaryEmbedding.rotate_queries_or_keys (rotary-embedding-torch 0.2.3; call site
    attention.py:709-711): interleaved pairs on the first 2*len(freqs) dims, seq dim = -2."""
    n = x.shape[-2]
    ang = (torch.arange(n, dtype=freqs.dtype)[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)
    rd = ang.shape[-1]
    xr, xp = x[..., :rd], x[..., rd:]
    x1, x2 = xr[..., 0::2], xr[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(xr.shape)
    return torch.cat([xr * ang.cos() + rot * ang.sin(), xp], dim=-1)


def temporal_attention(x, p, heads):
    """TemporalAttention.forward/_attention (attention.py:644-733); x: (B*HW, T, C)."""
    q = F.linear(x, p["to_q.weight"]); k = F.linear(x, p["to_k.weight"]); v = F.linear(x, p["to_v.weight"])
    d = q.shape[-1] // heads
    q = _heads(q, heads) * d ** -0.5
    k = _heads(k, heads); v = _heads(v, heads)
    freqs = p["rotary_emb.freqs"]
    q = rotary(q, freqs); k = rotary(k, freqs)
    s = q @ k.transpose(-1, -2) + time_rel_pos_bias(p["time_rel_pos_bias.relative_attention_bias.weight"], x.shape[1])
    s = s - s.amax(dim=-1, keepdim=True)
    o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(x.shape)
    return F.linear(o, p["to_out.0.weight"], p["to_out.0.bias"])


def feed_forward(x, p):
    """FeedForward with GEGLU (diffusers 0.16.0; spec copy diffusers_attention.py:735-823)."""
    h, gate = F.linear(x, p["net.0.proj.weight"], p["net.0.proj.bias"]).chunk(2, dim=-1)
    return F.linear(h * F.gelu(gate), p["net.2.weight"], p["net.2.bias"])


def layer_norm(x, p):
    return F.layer_norm(x, (x.shape[-1],), p["weight"], p["bias"], 1e-5)


def basic_transformer_block(x, ctx, p, heads, video_length, only_cross):
    """BasicTransformerBlock.forward (attention.py:523-564); x: (B*T, HW, C), ctx: (B*T, 77, Cx)."""
    n = layer_norm(x, p.sub("norm1"))
    x = cross_attention(n, ctx if only_cross else None, p.sub("attn1"), heads) + x
    n = layer_norm(x, p.sub("norm2"))
    x = cross_attention(n, ctx, p.sub("attn2"), heads) + x
    bt, d, c = x.shape
    b = bt // video_length
    xt = x.reshape(b, video_length, d, c).permute(0, 2, 1, 3).reshape(b * d, video_length, c)
    n = layer_norm(xt, p.sub("norm_temporal"))
    xt = temporal_attention(n, p.sub("attn_temporal"), heads) + xt
    x = xt.reshape(b, d, video_length, c).permute(0, 2, 1, 3).reshape(bt, d, c)
    return feed_forward(layer_norm(x, p.sub("norm3")), p.sub("ff")) + x


def transformer3d(x, ehs, p, heads, only_cross, groups=32):
    """Transformer3DModel.forward (attention.py:359-411), use_linear_projection=True."""
    b, c, t, h, w = x.shape
    x = resnet_block3d_cnn(x, p.sub("resblock_temporal"), None, kt=3)
    xf = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    res = xf
    n = F.group_norm(xf, groups, p["norm.weight"], p["norm.bias"], 1e-6)
    tok = n.permute(0, 2, 3, 1).reshape(b * t, h * w, c)
    tok = F.linear(tok, p["proj_in.weight"], p["proj_in.bias"])
    ctx = ehs.repeat_interleave(t, dim=0)             # repeat 'b n c -> (b f) n c' (attention.py:364)
    tok = basic_transformer_block(tok, ctx, p.sub("transformer_blocks.0"), heads, t, only_cross)
    tok = F.linear(tok, p["proj_out.weight"], p["proj_out.bias"])
    out = tok.reshape(b * t, h, w, c).permute(0, 3, 1, 2) + res
    return out.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def temporal_module3d(x, p, temb):
    """TemporalModule3D.forward (temporal_module.py:175-194), attention_block_types ("","")."""
    h = resnet_block3d_cnn(x, p.sub("resblocks_3d_temporal"), temb, kt=5)
    h = resnet_block3d(h, p.sub("resblocks_3d_spatial"), temb, 32, 32, 1e-6)
    h = conv2d_frames(h, p.sub("shift_conv"), padding=0)
    return x + h


def timestep_embedding(t, dim, flip_sin_to_cos=True, shift=0.0):
    """diffusers 0.16.0 get_timestep_embedding (call site unet_video.py:173,472)."""
    half = dim // 2
    expo = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - shift)
    emb = t[:, None].float() * torch.exp(expo)[None, :]
    emb = torch.cat([emb.sin(), emb.cos()], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


# ---------------------------------------------------------------------------------------------
def unet_forward(sd, cfg, sample, timestep, low_res, ehs, class_labels):
    """UNetVideoModel.forward (unet_video.py:404-574).  sample (B,4,T,H,W), low_res (B,3,T,H,W),
    ehs (B,77,Cx), timestep scalar, class_labels (1,) or (B,) long.  Returns (B,4,T,H,W)."""
    p = P(sd)
    boc = list(cfg["block_out_channels"])
    nblk = len(boc)
    hd = cfg["attention_head_dim"]
    hd = list(hd) if isinstance(hd, (list, tuple)) else [hd] * nblk
    oca = cfg.get("only_cross_attention", [True, True, True, False])
    oca = list(oca) if isinstance(oca, (list, tuple)) else [oca] * nblk
    lpb = cfg.get("layers_per_block", 2)
    eps = cfg.get("norm_eps", 1e-5)
    groups = cfg.get("norm_num_groups", 32)
    down_types = cfg["down_block_types"]; up_types = cfg["up_block_types"]
    down_t = set(cfg.get("down_temporal_idx", (0, 1, 2)))
    up_t = set(cfg.get("up_temporal_idx", (1, 2, 3)))
    mid_t = cfg.get("mid_temporal", False)
    osf_mid = cfg.get("mid_block_scale_factor", 1)

    x = torch.cat([sample, low_res], dim=1)                                     # :440
    bsz = x.shape[0]
    ts = torch.as_tensor(timestep).reshape(-1).expand(bsz)                      # :457-470
    temb = timestep_embedding(ts, boc[0], cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0))
    te = p.sub("time_embedding")
    emb = F.linear(F.silu(F.linear(temb, te["linear_1.weight"], te["linear_1.bias"])), te["linear_2.weight"], te["linear_2.bias"])
    emb = emb + p["class_embedding.weight"][class_labels.reshape(-1)]           # :480-491 (broadcast over batch)

    x = conv2d_frames(x, p.sub("conv_in"))                                      # :495
    skips = [x]
    for i, btype in enumerate(down_types):                                      # :499-518
        bp = p.sub(f"down_blocks.{i}")
        for j in range(lpb):
            x = resnet_block3d(x, bp.sub(f"resnets.{j}"), emb, groups, groups, eps)
            if btype == "CrossAttnDownBlock3D":
                x = transformer3d(x, ehs, bp.sub(f"attentions.{j}"), hd[i], oca[i], groups)
            skips.append(x)
        if i != nblk - 1:
            x = conv2d_frames(x, bp.sub("downsamplers.0.conv"), stride=2, padding=cfg.get("downsample_padding", 1))
            skips.append(x)
        if i in down_t:
            x = temporal_module3d(x, p.sub(f"down_temp_blocks.{i}"), emb)
    mp = p.sub("mid_block")                                                     # :522-531
    x = resnet_block3d(x, mp.sub("resnets.0"), emb, groups, groups, eps, osf_mid)
    x = transformer3d(x, ehs, mp.sub("attentions.0"), hd[-1], False, groups)
    x = resnet_block3d(x, mp.sub("resnets.1"), emb, groups, groups, eps, osf_mid)
    if mid_t:
        x = temporal_module3d(x, p.sub("mid_temp_block"), emb)
    rhd = list(reversed(hd)); roca = list(reversed(oca))
    for i, btype in enumerate(up_types):                                        # :533-564
        bp = p.sub(f"up_blocks.{i}")
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block3d(x, bp.sub(f"resnets.{j}"), emb, groups, groups, eps)
            if btype == "CrossAttnUpBlock3D":
                x = transformer3d(x, ehs, bp.sub(f"attentions.{j}"), rhd[i], roca[i], groups)
        if i != nblk - 1:
            if skips and skips[-1].shape[-2:] != (x.shape[-2] * 2, x.shape[-1] * 2):
                size = (x.shape[2],) + tuple(skips[-1].shape[-2:])              # forward_upsample_size :541-542
                x = F.interpolate(x, size=size, mode="nearest")
            else:
                x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
            x = conv2d_frames(x, bp.sub("upsamplers.0.conv"))
        if i in up_t:
            x = temporal_module3d(x, p.sub(f"up_temp_blocks.{i}"), emb)
    x = F.silu(group_norm(x, p.sub("conv_norm_out"), groups, eps))              # :567-569
    return conv2d_frames(x, p.sub("conv_out"))


# ---------------------------------------------------------------------------------------------
def vae_attention_block(x, p, groups=32, eps=1e-6, chunk=4096):
    """diffusers 0.16.0 AttentionBlock (spec copy diffusers_attention.py:331-381), single head,
    applied per frame (unet_blocks.py:740-742).  x: (N,C,H,W).  Row-chunked softmax =
    mathematically identical to baddbmm+softmax+bmm, bounded memory."""
    n, c, h, w = x.shape
    res = x
    t = F.group_norm(x, groups, p["group_norm.weight"], p["group_norm.bias"], eps)
    t = t.reshape(n, c, h * w).transpose(1, 2)
    q = F.linear(t, p["query.weight"], p["query.bias"])
    k = F.linear(t, p["key.weight"], p["key.bias"])
    v = F.linear(t, p["value.weight"], p["value.bias"])
    scale = 1.0 / math.sqrt(c)
    out = torch.empty_like(q)
    for s in range(0, h * w, chunk):
        a = torch.softmax(q[:, s:s + chunk] @ k.transpose(1, 2) * scale, dim=-1)
        out[:, s:s + chunk] = a @ v
    out = F.linear(out, p["proj_attn.weight"], p["proj_attn.bias"])
    return out.transpose(1, 2).reshape(n, c, h, w) + res


def vae_decode(sd, cfg, z, img=None, w_lr=1.0, pre_clamp=True):
    """AutoencoderKLVideo.decode/_decode_cond (autoencoder_kl_cond_video.py:199-226) ->
    Decoder.forward (vae_video.py:365-405).  z: (B,4,T,H,W) already divided by scaling_factor."""
    p = P(sd)
    groups = cfg.get("norm_num_groups", 32)
    plus = cfg["up_block_types"][0] == "UpDecoderBlock3D_plus"
    res = resnet_block3d_plus if plus else (lambda x, pp, groups=32, groups_out=None, eps=1e-6:
                                            resnet_block3d(x, pp, None, groups, groups_out, eps))
    lpb = cfg.get("layers_per_block", 2)
    x = conv2d_frames(z, p.sub("post_quant_conv"), padding=0)
    d = p.sub("decoder")
    x = conv2d_frames(x, d.sub("conv_in"))
    if cfg.get("condition_img", False):                                         # vae_video.py:370-373
        c = resnet_block3d_plus(img, d.sub("condition_in.0"), groups=3, groups_out=32)
        c = resnet_block3d_plus(c, d.sub("condition_in.1"))
        f = d.sub("condition_fuse")                                             # Fuse_sft_block resnet.py:73-79
        e = resnet_block3d(torch.cat([c, x], dim=1), f.sub("shared.0"))
        e = resnet_block3d(e, f.sub("shared.1"))
        x = x + w_lr * (x * conv2d_frames(e, f.sub("scale")) + conv2d_frames(e, f.sub("shift")))
    m = d.sub("mid_block")                                                      # unet_blocks.py:735-745
    x = res(x, m.sub("resnets.0"), groups)
    b, c_, t, h, w = x.shape
    xa = vae_attention_block(x.permute(0, 2, 1, 3, 4).reshape(b * t, c_, h, w), m.sub("attentions.0"), groups)
    x = xa.reshape(b, t, c_, h, w).permute(0, 2, 1, 3, 4)
    x = res(x, m.sub("resnets.1"), groups)
    nup = len(cfg["up_block_types"])
    for i in range(nup):                                                        # unet_blocks.py:851-859
        u = d.sub(f"up_blocks.{i}")
        for j in range(lpb + 1):
            x = res(x, u.sub(f"resnets.{j}"), groups)
        if i != nup - 1:
            x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
            x = conv2d_frames(x, u.sub("upsamplers.0.conv"))
    x = F.silu(group_norm(x, d.sub("conv_norm_out"), groups, 1e-6))
    return conv2d_frames(x, d.sub("conv_out"))


# ---------------------------------------------------------------------------------------------
class DDIM:
    """DDIMScheduler split into step_v0 / step_vt (scheduling_ddim.py:130-176,237-259,383-520).
    Defaults = upstream SD-x4-upscaler scheduler_config.json (absent from the reference tree;
    SURVEY.md §8c): scaled_linear 0.00085..0.012, v_prediction, steps_offset 1,
    set_alpha_to_one False, clip_sample False."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="v_prediction",
                 clip_sample_range=1.0):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.n_train = num_train_timesteps
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        self.clip_sample, self.clip_range = clip_sample, clip_sample_range
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n_inf = n
        ratio = self.n_train // n
        self.timesteps = [int(round(i * ratio)) + self.steps_offset for i in range(n)][::-1]
        return self.timesteps

    def _alphas(self, t):
        prev = t - self.n_train // self.n_inf
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step_v0(self, model_output, t, sample):
        a_t, _ = self._alphas(t)
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        elif self.prediction_type == "sample":
            x0 = model_output
        else:
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_range, self.clip_range)
        return x0

    def step_vt(self, x0, model_output, t, sample):
        a_t, a_prev = self._alphas(t)
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            eps = model_output
        elif self.prediction_type == "sample":
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        else:
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_range, self.clip_range)
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps        # eta = 0


def add_noise(x, noise, level, beta_start=0.0001, beta_end=0.02, n=1000):
    """low_res_scheduler.add_noise (DDPMScheduler, scaled_linear; math scheduling_ddim.py:524-545)."""
    # device-explicit (the GPU shim's `torch.device` context was seen NOT to reach this factory call in a long pytest process)
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32, device=x.device) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)[level.to(x.device)]
    return ac ** 0.5 * x + (1 - ac) ** 0.5 * noise


# ---------------------------------------------------------------------------------------------
def flow_warp(x, flow, mode="bilinear"):
    """flow_warp (propagation_module.py:104-135); flow: (n,h,w,2)."""
    _, _, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(0, h).type_as(x), torch.arange(0, w).type_as(x), indexing="ij")
    vx = gx + flow[..., 0]; vy = gy + flow[..., 1]
    grid = torch.stack((2.0 * vx / max(w - 1, 1) - 1.0, 2.0 * vy / max(h - 1, 1) - 1.0), dim=3)
    return F.grid_sample(x, grid, mode=mode, padding_mode="zeros", align_corners=True)


def fb_consistency(flow_fw, flow_bw, a1, a2):
    """fbConsistencyCheck (propagation_module.py:140-149)."""
    bw = flow_warp(flow_bw, flow_fw.permute(0, 2, 3, 1))
    lsq = lambda t: (t * t).sum(dim=1, keepdim=True)
    return (lsq(flow_fw + bw) < a1 * (lsq(flow_fw) + lsq(bw)) + a2).to(flow_fw)


def propagation(x, flows_f, flows_b, interpolation="nearest", fuse_scale=0.5, a1=0.001, a2=0.05):
    """Propagation.forward, learnable=False, mode='fuse' (propagation_module.py:194-281)."""
    b, c, t, h, w = x.shape
    s = 1.0 * w / flows_f.shape[-1]
    flows_f = F.interpolate(flows_f, (t - 1, h, w), mode="area") * s
    flows_b = F.interpolate(flows_b, (t - 1, h, w), mode="area") * s
    feats = [x[:, :, i] for i in range(t)]
    for direction in ("backward", "forward"):
        if direction == "backward":
            order = list(range(t))[::-1]; fidx = order; fprop, fchk = flows_f, flows_b
        else:
            order = list(range(t)); fidx = list(range(-1, t - 1)); fprop, fchk = flows_b, flows_f
        out = []
        prop = None
        for i, idx in enumerate(order):
            cur = feats[idx]
            if i == 0:
                prop = cur
            else:
                fp = fprop[:, :, fidx[i]]; fc = fchk[:, :, fidx[i]]
                mask = fb_consistency(fp, fc, a1, a2)
                warped = flow_warp(prop, fp.permute(0, 2, 3, 1), interpolation)
                warped = warped * fuse_scale + cur * (1 - fuse_scale)
                prop = mask * warped + (1 - mask) * cur
            out.append(prop)
        feats = out[::-1] if direction == "backward" else out
    return torch.stack(feats, dim=2)


# ---------------------------------------------------------------------------------------------
def window_schedule(t_total, short_seq=8, overlap=2):
    """Temporal windows of the denoising loop (pipeline_upscale_a_video.py:601-629), in visiting
    order, INCLUDING the reference's duplicate final window."""
    if t_total <= short_seq:
        return [(0, t_total)]
    out = []
    for s in range(0, t_total, short_seq - overlap):
        e = min(t_total, s + short_seq)
        if e - s < short_seq:
            s = e - short_seq
        out.append((s, e))
    return out


def pipeline_call(unet_sd, unet_cfg, vae_sd, vae_cfg, image, prompt_embeds, *, num_inference_steps, guidance_scale,
                  noise_level, lr_noise, latents, flows_bi=None, propagation_steps=(), w_lr=1.0, scheduler_kwargs=None,
                  decode=True, return_trace=False, resume=None):
    """VideoUpscalePipeline.__call__ (pipeline_upscale_a_video.py:436-716) in fp32.
    image (1,3,T,h,w) in [-1,1]; prompt_embeds (2,77,C) = [negative, positive]; lr_noise and
    latents are the two randn draws of the reference (:547, :567), injected so that RNG order does
    not matter.  Returns (images (1,3,T,4h,4w) clamped, latents_out).
    `resume=(i0, lat)` (test economy): start the loop at step index i0 from the latents a previous call traced after step
    i0 - 1 — the steps before a first propagation step are shared by a run with and a run without propagation."""
    sch = DDIM(**(scheduler_kwargs or {}))
    do_cfg = guidance_scale > 1.0
    image_dec = image.clone()
    nl = torch.tensor([noise_level], dtype=torch.long)
    img = add_noise(image, lr_noise, nl)                                        # :546-548
    img = torch.cat([img] * 2) if do_cfg else img                               # :550-551
    t_total = img.shape[2]
    timesteps = sch.set_timesteps(num_inference_steps)
    lat = latents * sch.init_noise_sigma
    trace = []
    if resume is not None:
        lat = resume[1].clone()
    for i, t in enumerate(timesteps):                                           # :607
        if resume is not None and i < resume[0]:
            continue
        lin = torch.cat([lat] * 2) if do_cfg else lat
        wins = window_schedule(t_total)
        if len(wins) > 1:                                                       # :619-635
            cl = torch.cat([nl] * img.shape[0])
            preds = [None] * t_total
            for (s, e) in wins:
                o = unet_forward(unet_sd, unet_cfg, lin[:, :, s:e], t, img[:, :, s:e], prompt_embeds, cl)
                for kk, idx in enumerate(range(s, e)):
                    preds[idx] = o[:, :, kk:kk + 1] if preds[idx] is None else preds[idx] * 0.5 + o[:, :, kk:kk + 1] * 0.5
            eps = torch.cat(preds, dim=2)
        else:
            eps = unet_forward(unet_sd, unet_cfg, lin, t, img, prompt_embeds, nl)   # :637-639
        if do_cfg:
            eu, ec = eps.chunk(2)
            eps = eu + guidance_scale * (ec - eu)                               # :643-645
        x0 = sch.step_v0(eps, t, lat)                                           # :649
        if flows_bi is not None and i in propagation_steps:                     # :652-657
            x0 = propagation(x0, flows_bi[0], flows_bi[1], "nearest", 0.5, 0.001, 0.05)
        lat = sch.step_vt(x0, eps, t, lat)                                      # :659
        if return_trace:
            trace.append(lat.clone())
    latents_out = lat.clone()
    if not decode:
        return (None, latents_out, trace) if return_trace else (None, latents_out)
    sf = vae_cfg.get("scaling_factor", 0.08333)
    outs = []
    for s in range(0, t_total, 3) if t_total > 3 else [0]:                      # :685-702
        e = min(t_total, s + 3) if t_total > 3 else t_total
        dec = vae_decode(vae_sd, vae_cfg, lat[:, :, s:e] / sf, image_dec[:, :, s:e], w_lr)
        outs.append(dec.clamp(-1, 1))
    images = torch.cat(outs, dim=2)
    return (images, latents_out, trace) if return_trace else (images, latents_out)


# ---------------------------------------------------------------------------------------------
# RAFT (models_video/RAFT/*.py) — fp32, "things" configuration (non-small, no alternate corr)
def _norm2d(x, p, kind):
    """norm layers of extractor.py: InstanceNorm2d (no affine, eps 1e-5) for fnet, eval-mode
    BatchNorm2d (running stats) for cnet."""
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, p["running_mean"], p["running_var"], p["weight"], p["bias"], False, 0.0, 1e-5)


def _raft_resblock(x, p, kind, stride):
    """ResidualBlock.forward (extractor.py:47-55)."""
    y = F.relu(_norm2d(F.conv2d(x, p["conv1.weight"], p["conv1.bias"], stride, 1), p.sub("norm1"), kind))
    y = F.relu(_norm2d(F.conv2d(y, p["conv2.weight"], p["conv2.bias"], 1, 1), p.sub("norm2"), kind))
    if stride != 1:
        x = _norm2d(F.conv2d(x, p["downsample.0.weight"], p["downsample.0.bias"], stride), p.sub("downsample.1"), kind)
    return F.relu(x + y)


def raft_encoder(x, p, kind):
    """BasicEncoder.forward (extractor.py:166-193)."""
    x = F.relu(_norm2d(F.conv2d(x, p["conv1.weight"], p["conv1.bias"], 2, 3), p.sub("norm1"), kind))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _raft_resblock(x, p.sub(name + ".0"), kind, stride)
        x = _raft_resblock(x, p.sub(name + ".1"), kind, 1)
    return F.conv2d(x, p["conv2.weight"], p["conv2.bias"])


def _bilinear_sampler(img, coords):
    """utils/utils.py:57-71 (pixel coordinates, align_corners=True, zero padding)."""
    h, w = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    grid = torch.cat([2 * xg / (w - 1) - 1, 2 * yg / (h - 1) - 1], dim=-1)
    return F.grid_sample(img, grid, align_corners=True)


def raft_corr_pyramid(f1, f2, levels=4):
    """CorrBlock.__init__ / corr (corr.py:12-27,53-60)."""
    b, d, h, w = f1.shape
    corr = (f1.view(b, d, h * w).transpose(1, 2) @ f2.view(b, d, h * w)) / torch.sqrt(torch.tensor(d).float())
    corr = corr.reshape(b * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def raft_corr_lookup(pyr, coords, r=4):
    """CorrBlock.__call__ (corr.py:29-50)."""
    coords = coords.permute(0, 2, 3, 1)
    b, h, w, _ = coords.shape
    out = []
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)        # (dy-index, dx-index, 2) added to (x, y)!
    for i, corr in enumerate(pyr):
        cl = coords.reshape(b * h * w, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        out.append(_bilinear_sampler(corr, cl).view(b, h, w, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def raft_update(net, inp, corr, flow, p):
    """BasicUpdateBlock.forward (update.py:129-139) with BasicMotionEncoder (:94-103), SepConvGRU (:44-60),
    FlowHead (:13-14).  Returns (net, delta_flow); the mask head is evaluated by the caller."""
    e = p.sub("encoder")
    cor = F.relu(F.conv2d(corr, e["convc1.weight"], e["convc1.bias"]))
    cor = F.relu(F.conv2d(cor, e["convc2.weight"], e["convc2.bias"], padding=1))
    flo = F.relu(F.conv2d(flow, e["convf1.weight"], e["convf1.bias"], padding=3))
    flo = F.relu(F.conv2d(flo, e["convf2.weight"], e["convf2.bias"], padding=1))
    out = F.relu(F.conv2d(torch.cat([cor, flo], 1), e["conv.weight"], e["conv.bias"], padding=1))
    x = torch.cat([inp, out, flow], dim=1)
    g = p.sub("gru")
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([net, x], dim=1)
        z = torch.sigmoid(F.conv2d(hx, g["convz" + sfx + ".weight"], g["convz" + sfx + ".bias"], padding=pad))
        r = torch.sigmoid(F.conv2d(hx, g["convr" + sfx + ".weight"], g["convr" + sfx + ".bias"], padding=pad))
        q = torch.tanh(F.conv2d(torch.cat([r * net, x], 1), g["convq" + sfx + ".weight"], g["convq" + sfx + ".bias"], padding=pad))
        net = (1 - z) * net + z * q
    fh = p.sub("flow_head")
    delta = F.conv2d(F.relu(F.conv2d(net, fh["conv1.weight"], fh["conv1.bias"], padding=1)), fh["conv2.weight"], fh["conv2.bias"], padding=1)
    return net, delta


def raft_upsample_flow(flow, mask):
    """RAFT.upsample_flow (raft.py:73-85): convex combination over the 3x3 neighbourhood, x8."""
    n, _, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


def raft_forward(sd, image1, image2, iters=20):
    """RAFT.forward(test_mode=True) (raft.py:87-145) -> upsampled flow (N,2,H,W)."""
    p = P(sd)
    f = raft_encoder(torch.cat([image1, image2], 0), p.sub("fnet"), "instance").float()
    f1, f2 = f.split(image1.shape[0], 0)
    pyr = raft_corr_pyramid(f1, f2)
    c = raft_encoder(image1, p.sub("cnet"), "batch")
    net, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
    n, _, h, w = image1.shape
    ys, xs = torch.meshgrid(torch.arange(h // 8), torch.arange(w // 8), indexing="ij")
    coords0 = torch.stack([xs, ys], 0).float()[None].repeat(n, 1, 1, 1)
    coords1 = coords0.clone()
    u = p.sub("update_block")
    for _ in range(iters):
        corr = raft_corr_lookup(pyr, coords1)
        net, delta = raft_update(net, inp, corr, coords1 - coords0, u)
        coords1 = coords1 + delta
    m = u.sub("mask")
    mask = 0.25 * F.conv2d(F.relu(F.conv2d(net, m["0.weight"], m["0.bias"], padding=1)), m["2.weight"], m["2.bias"])
    return raft_upsample_flow(coords1 - coords0, mask)


def raft_bi_forward(sd, frames, iters=20):
    """RAFT_bi.forward (raft_bi.py:47-68).  frames (B,3,T,H,W) -> flows fwd/bwd (B,2,T-1,H,W).
    H, W not multiples of 8: trilinear pre-resize to the next multiple (:49-53; T is unchanged, so per-frame
    bilinear) and resize_flow_pytorch back (:11-16,:62-63) — including the reference's rescale of
    `flow[:, :, 0]` / `flow[:, :, 1]`, which indexes ROWS 0 and 1 of both flow channels."""
    b, c, t, h, w = frames.shape
    h8, w8 = -(-h // 8) * 8, -(-w // 8) * 8
    if (h8, w8) != (h, w):
        frames = F.interpolate(frames, (t, h8, w8), mode="trilinear")
    a = frames[:, :, :-1].permute(0, 2, 1, 3, 4).reshape(b * (t - 1), c, h8, w8)
    bb = frames[:, :, 1:].permute(0, 2, 1, 3, 4).reshape(b * (t - 1), c, h8, w8)
    ff = raft_forward(sd, a, bb, iters)
    fb = raft_forward(sd, bb, a, iters)
    if (h8, w8) != (h, w):
        def back(flow):
            flow = F.interpolate(flow, (h, w), mode="bilinear")
            flow[:, :, 0] *= h / h8
            flow[:, :, 1] *= w / w8
            return flow
        ff, fb = back(ff), back(fb)
    ff = ff.reshape(b, t - 1, 2, h, w).permute(0, 2, 1, 3, 4)
    fb = fb.reshape(b, t - 1, 2, h, w).permute(0, 2, 1, 3, 4)
    return ff.contiguous(), fb.contiguous()


# ---------------------------------------------------------------------------------------------
# colour correction (models_video/color_correction.py), fp32
def adain(content, style, eps=1e-5):
    """adaptive_instance_normalization (:59-71) with calc_mean_std (:43-57: UNBIASED variance + eps)."""
    def ms(f):
        b, c = f.shape[:2]
        flat = f.reshape(b, c, -1)
        return flat.mean(dim=2).reshape(b, c, 1, 1), (flat.var(dim=2) + eps).sqrt().reshape(b, c, 1, 1)
    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm


def atrous_blur(image, radius):
    """wavelet_blur (:73-91): [1 2 1]x[1 2 1]/16 with dilation `radius` on a replicate-padded image."""
    k1 = torch.tensor([0.25, 0.5, 0.25], dtype=image.dtype)
    k = (k1[:, None] * k1[None, :])[None, None].repeat(image.shape[1], 1, 1, 1)
    return F.conv2d(F.pad(image, (radius,) * 4, mode="replicate"), k, groups=image.shape[1], dilation=radius)


def wavelet_decomposition(image, levels=5):
    high = torch.zeros_like(image)                                              # :93-106
    for i in range(levels):
        low = atrous_blur(image, 2 ** i)
        high = high + (image - low)
        image = low
    return high, image


def wavelet_reconstruction(content, style):
    return wavelet_decomposition(content)[0] + wavelet_decomposition(style)[1]  # :108-118


def bicubic4(frames):
    return F.interpolate(frames, scale_factor=4, mode="bicubic")                 # inference_upscale_a_video.py:327
