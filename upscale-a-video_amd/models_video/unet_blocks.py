"""MI355X-native UNet / VAE blocks (drop-in for the reference's `models_video/unet_blocks.py`:
get_down_block :19, get_up_block :94, UNetMidBlock3DCrossAttn :180, CrossAttnDownBlock3D :270,
DownBlock3D :391, CrossAttnUpBlock3D :470, UpBlock3D :593, UNetMidBlock3D(_plus) :666/:862,
UpDecoderBlock3D(_plus) :808/:943).

The blocks only sequence channels-last `run(...)` calls of the resnet / transformer modules; the
skip-connection concat of the up blocks (reference :563,:646 `torch.cat`) is never materialised:
GroupNorm statistics / apply and the 1x1 shortcut conv read the two tensors directly.
"""
import torch
import torch.nn as nn

from uav import engine as E
from uav import ops

from .attention import Transformer3DModel
from .resnet import Downsample3D, ResnetBlock3D, ResnetBlock3D_plus, Upsample3D


def _res_kwargs(eps, groups, osf, act, tnorm, pre_norm, dropout):
    return dict(eps=eps, groups=groups, dropout=dropout, time_embedding_norm=tnorm, non_linearity=act,
                output_scale_factor=osf, pre_norm=pre_norm)


def _tf(heads, channels, cross_attention_dim, groups, use_linear_projection, only_cross_attention, upcast_attention,
        use_first_frame, use_relative_position, rotary_emb):
    return Transformer3DModel(heads, channels // heads, in_channels=channels, num_layers=1,
                              cross_attention_dim=cross_attention_dim, norm_num_groups=groups,
                              use_linear_projection=use_linear_projection, only_cross_attention=only_cross_attention,
                              upcast_attention=upcast_attention, use_first_frame=use_first_frame,
                              use_relative_position=use_relative_position, rotary_emb=rotary_emb)


class UNetMidBlock3DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 attn_num_head_channels=1, output_scale_factor=1.0, cross_attention_dim=1280, dual_cross_attention=False,
                 use_linear_projection=False, upcast_attention=False, use_first_frame=False, use_relative_position=False,
                 rotary_emb=None):
        super().__init__()
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        resnet_groups = resnet_groups if resnet_groups is not None else min(in_channels // 4, 32)
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        resnets = [ResnetBlock3D(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels, **kw)]
        attentions = []
        for _ in range(num_layers):
            attentions.append(_tf(attn_num_head_channels, in_channels, cross_attention_dim, resnet_groups,
                                  use_linear_projection, False, upcast_attention, use_first_frame,
                                  use_relative_position, rotary_emb))
            resnets.append(ResnetBlock3D(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels, **kw))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)

    def run(self, x, g, temb, ehs_rows, n_text):
        x = self.resnets[0].run(x, g, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            x = attn.run(x, g, ehs_rows, n_text)
            x = resnet.run(x, g, temb)
        return x


class CrossAttnDownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0, downsample_padding=1,
                 add_downsample=True, dual_cross_attention=False, use_linear_projection=False, only_cross_attention=False,
                 upcast_attention=False, use_first_frame=False, use_relative_position=False, rotary_emb=None):
        super().__init__()
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        resnets, attentions = [], []
        for i in range(num_layers):
            resnets.append(ResnetBlock3D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                                         temb_channels=temb_channels, **kw))
            attentions.append(_tf(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups,
                                  use_linear_projection, only_cross_attention, upcast_attention, use_first_frame,
                                  use_relative_position, rotary_emb))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None
        self.gradient_checkpointing = False

    def run(self, x, g, temb, ehs_rows, n_text):
        outs = []
        for resnet, attn in zip(self.resnets, self.attentions):
            x = resnet.run(x, g, temb)
            x = attn.run(x, g, ehs_rows, n_text)
            outs.append((x, g))
        if self.downsamplers is not None:
            x, g = self.downsamplers[0].run(x, g)
            outs.append((x, g))
        return x, g, outs


class DownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 output_scale_factor=1.0, add_downsample=True, downsample_padding=1):
        super().__init__()
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels, **kw)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None
        self.gradient_checkpointing = False
        self.has_cross_attention = False

    def run(self, x, g, temb, ehs_rows=None, n_text=0):
        outs = []
        for resnet in self.resnets:
            x = resnet.run(x, g, temb)
            outs.append((x, g))
        if self.downsamplers is not None:
            x, g = self.downsamplers[0].run(x, g)
            outs.append((x, g))
        return x, g, outs


def _target_size(skips, upsample_size):
    """Size the upsampler must produce: the next skip connection's (reference `upsample_size =
    down_block_res_samples[-1].shape[2:]`, unet_video.py:541-542).  Equal to 2x unless H or W is not a multiple of
    2^num_upsamplers, where the stride-2 convs rounded up on the way down."""
    if upsample_size is not None:
        return upsample_size
    if skips:
        gs = skips[-1][1]
        return (gs.h, gs.w)
    return None


class CrossAttnUpBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, dropout=0.0, num_layers=1,
                 resnet_eps=1e-6, resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32,
                 resnet_pre_norm=True, attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0,
                 add_upsample=True, dual_cross_attention=False, use_linear_projection=False, only_cross_attention=False,
                 upcast_attention=False, use_first_frame=False, use_relative_position=False, rotary_emb=None):
        super().__init__()
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        resnets, attentions = [], []
        for i in range(num_layers):
            skip = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock3D(in_channels=rin + skip, out_channels=out_channels, temb_channels=temb_channels, **kw))
            attentions.append(_tf(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups,
                                  use_linear_projection, only_cross_attention, upcast_attention, use_first_frame,
                                  use_relative_position, rotary_emb))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None
        self.gradient_checkpointing = False

    def run(self, x, g, skips, temb, ehs_rows, n_text, upsample_size=None):
        for resnet, attn in zip(self.resnets, self.attentions):
            skip, _ = skips.pop()
            x = resnet.run(x, g, temb, x2=skip)              # cat([hidden, skip], C) without the cat
            x = attn.run(x, g, ehs_rows, n_text)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0].run(x, g, _target_size(skips, upsample_size))
        return x, g


class UpBlock3D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, dropout=0.0, num_layers=1,
                 resnet_eps=1e-6, resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32,
                 resnet_pre_norm=True, output_scale_factor=1.0, add_upsample=True):
        super().__init__()
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        resnets = []
        for i in range(num_layers):
            skip = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock3D(in_channels=rin + skip, out_channels=out_channels, temb_channels=temb_channels, **kw))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None
        self.gradient_checkpointing = False

    def run(self, x, g, skips, temb, ehs_rows=None, n_text=0, upsample_size=None):
        for resnet in self.resnets:
            skip, _ = skips.pop()
            x = resnet.run(x, g, temb, x2=skip)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0].run(x, g, _target_size(skips, upsample_size))
        return x, g


# ---------------------------------------------------------------------------------------------
class AttentionBlock(E.EngineModule):
    """diffusers 0.16.0 AttentionBlock (un-vendored in the reference; spec copy
    models_video/diffusers_attention.py:249-381): per-frame GroupNorm -> q,k,v Linear (with bias)
    -> single-head softmax(QK^T/sqrt(C))V -> proj_attn -> + residual.  The L x L scores
    (L = H*W up to 102400) stay on chip in the flash kernel."""

    def __init__(self, channels, num_head_channels=None, norm_num_groups=32, rescale_output_factor=1.0, eps=1e-5):
        super().__init__()
        self.channels = channels
        self.num_heads = channels // num_head_channels if num_head_channels is not None else 1
        self.group_norm = nn.GroupNorm(num_channels=channels, num_groups=norm_num_groups, eps=eps, affine=True)
        self.query = nn.Linear(channels, channels)
        self.key = nn.Linear(channels, channels)
        self.value = nn.Linear(channels, channels)
        self.rescale_output_factor = rescale_output_factor
        self.proj_attn = nn.Linear(channels, channels, 1)
        self._use_memory_efficient_attention_xformers = False

    def run(self, x, g: E.Geom):
        c, d = self.channels, self.channels // self.num_heads
        n = E.group_norm(self, "gn", self.group_norm, x, n_inst=g.n_img, rows_per_inst=g.hw, silu=False)
        qkv = ops.linear(n, E.packed_cat(self, "qkv", [self.query, self.key, self.value]))
        o = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], bq=g.n_img, lq=g.hw, lk=g.hw, heads=self.num_heads,
                          head_dim=d, scale=1.0 / (d ** 0.5), q_stride=3 * c, k_stride=3 * c, v_stride=3 * c)
        return ops.linear(o, E.packed_conv(self, "proj", self.proj_attn), residual=x,
                          out_scale=1.0 / self.rescale_output_factor, out_f32=x.dtype == torch.float32,
                          gn_groups=self.group_norm.num_groups)


class _MidBase(nn.Module):
    RES = ResnetBlock3D

    def __init__(self, in_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 add_attention=True, attn_num_head_channels=1, output_scale_factor=1.0):
        super().__init__()
        resnet_groups = resnet_groups if resnet_groups is not None else min(in_channels // 4, 32)
        self.add_attention = add_attention
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        resnets = [self.RES(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels, **kw)]
        attentions = []
        for _ in range(num_layers):
            attentions.append(AttentionBlock(in_channels, num_head_channels=attn_num_head_channels,
                                             rescale_output_factor=output_scale_factor, eps=resnet_eps,
                                             norm_num_groups=resnet_groups) if add_attention else None)
            resnets.append(self.RES(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels, **kw))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)

    def run(self, x, g, temb=None):
        x = self.resnets[0].run(x, g, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            if attn is not None:
                x = attn.run(x, g)
            x = resnet.run(x, g, temb)
        return x


class UNetMidBlock3D(_MidBase):
    RES = ResnetBlock3D


class UNetMidBlock3D_plus(_MidBase):
    RES = ResnetBlock3D_plus


class _UpDecBase(nn.Module):
    RES = ResnetBlock3D

    def __init__(self, in_channels, out_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 output_scale_factor=1.0, add_upsample=True):
        super().__init__()
        kw = _res_kwargs(resnet_eps, resnet_groups, output_scale_factor, resnet_act_fn, resnet_time_scale_shift,
                         resnet_pre_norm, dropout)
        self.resnets = nn.ModuleList([self.RES(in_channels=in_channels if i == 0 else out_channels,
                                               out_channels=out_channels, temb_channels=None, **kw)
                                      for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

    def run(self, x, g):
        for resnet in self.resnets:
            x = resnet.run(x, g, None)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0].run(x, g)
        return x, g


class UpDecoderBlock3D(_UpDecBase):
    RES = ResnetBlock3D


class UpDecoderBlock3D_plus(_UpDecBase):
    RES = ResnetBlock3D_plus


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                   resnet_act_fn, attn_num_head_channels, resnet_groups=None, cross_attention_dim=None,
                   downsample_padding=None, dual_cross_attention=False, use_linear_projection=False,
                   only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default",
                   use_first_frame=False, use_relative_position=False, rotary_emb=None):
    down_block_type = down_block_type[7:] if down_block_type.startswith("UNetRes") else down_block_type
    if down_block_type == "DownBlock3D":
        return DownBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                           temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
                           resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups, downsample_padding=downsample_padding,
                           resnet_time_scale_shift=resnet_time_scale_shift)
    if down_block_type == "CrossAttnDownBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock3D")
        return CrossAttnDownBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                                    temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
                                    resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups,
                                    downsample_padding=downsample_padding, cross_attention_dim=cross_attention_dim,
                                    attn_num_head_channels=attn_num_head_channels, dual_cross_attention=dual_cross_attention,
                                    use_linear_projection=use_linear_projection, only_cross_attention=only_cross_attention,
                                    upcast_attention=upcast_attention, resnet_time_scale_shift=resnet_time_scale_shift,
                                    use_first_frame=use_first_frame, use_relative_position=use_relative_position,
                                    rotary_emb=rotary_emb)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 resnet_eps, resnet_act_fn, attn_num_head_channels, resnet_groups=None, cross_attention_dim=None,
                 dual_cross_attention=False, use_linear_projection=False, only_cross_attention=False,
                 upcast_attention=False, resnet_time_scale_shift="default", use_first_frame=False,
                 use_relative_position=False, rotary_emb=None):
    up_block_type = up_block_type[7:] if up_block_type.startswith("UNetRes") else up_block_type
    if up_block_type == "UpBlock3D":
        return UpBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                         prev_output_channel=prev_output_channel, temb_channels=temb_channels, add_upsample=add_upsample,
                         resnet_eps=resnet_eps, resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups,
                         resnet_time_scale_shift=resnet_time_scale_shift)
    if up_block_type == "CrossAttnUpBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlock3D")
        return CrossAttnUpBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                                  prev_output_channel=prev_output_channel, temb_channels=temb_channels,
                                  add_upsample=add_upsample, resnet_eps=resnet_eps, resnet_act_fn=resnet_act_fn,
                                  resnet_groups=resnet_groups, cross_attention_dim=cross_attention_dim,
                                  attn_num_head_channels=attn_num_head_channels, dual_cross_attention=dual_cross_attention,
                                  use_linear_projection=use_linear_projection, only_cross_attention=only_cross_attention,
                                  upcast_attention=upcast_attention, resnet_time_scale_shift=resnet_time_scale_shift,
                                  use_first_frame=use_first_frame, use_relative_position=use_relative_position,
                                  rotary_emb=rotary_emb)
    if up_block_type == "UpDecoderBlock3D":
        return UpDecoderBlock3D(in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                                resnet_eps=resnet_eps, resnet_time_scale_shift=resnet_time_scale_shift,
                                resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups, add_upsample=add_upsample)
    if up_block_type == "UpDecoderBlock3D_plus":
        return UpDecoderBlock3D_plus(in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                                     resnet_eps=resnet_eps, resnet_time_scale_shift=resnet_time_scale_shift,
                                     resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups, add_upsample=add_upsample)
    raise ValueError(f"{up_block_type} does not exist.")
