"""MI355X-native ResNet / conv blocks of Upscale-A-Video (drop-in for the reference's
`models_video/resnet.py`).

Same class names, constructor arguments and state-dict keys as the reference
(/root/reference/models_video/resnet.py: InflatedConv3d :94, Upsample3D :104, Downsample3D :161,
ResnetBlock3D :200, ResnetBlock3DCNN :297, ResnetBlock3D_plus :396, Fuse_sft_block :63); the
compute is the HIP kernel library libuav_hip.so:

  GroupNorm(5-D statistics) -> SiLU       uav_groupnorm_scale_shift + uav_groupnorm_apply
  3x3 / (k,1,1) / 3x3x3 / 1x1 conv        uav_conv_gemm_f16 (implicit GEMM on MFMA) with the
                                          bias, time-embedding add (:272-276), residual /
                                          shortcut add and 1/output_scale_factor (:292) fused in
                                          the epilogue, nearest-2x upsampling folded into the
                                          gather (:144), skip-concat read from two tensors.

Modules run on channels-last fp16 rows `[B*T*H*W][C]` (`run(...)`); `forward(...)` keeps the
reference's (B,C,T,H,W) tensor signature for API parity and converts at the edge.
"""
import os

import torch
import torch.nn as nn

from uav import engine as E
from uav import ops


# "nearest 2x + 3x3 conv" as four 2x2 sub-pixel phase convs (2.25x fewer multiply-adds; ops.upsample_phase_weights).
# UAV_PHASE_UPSAMPLE=0 keeps the fused-gather form (upsampling folded into the 3x3 conv's addressing).
PHASE_UPSAMPLE = os.environ.get("UAV_PHASE_UPSAMPLE", "1") != "0"
# GroupNorm statistics of an up-sampled tensor from its four phase launches (shared workspace; 0: stand-alone statistics pass)
FUSE_UPSAMPLE_GN = os.environ.get("UAV_FUSE_UPSAMPLE_GN", "1") != "0"


def _temb_rows(mod, temb):
    """time_emb_proj(SiLU(temb)) -> fp32 [B][Cout] (resnet.py:272-273)."""
    lin = mod.time_emb_proj
    w = E.f16_param(mod, "time_emb_proj.w", lin.weight)
    b = E.f32_param(mod, "time_emb_proj.b", lin.bias)
    return ops.linear_small(temb, w, b, pre_silu=True)


class InflatedConv3d(nn.Conv2d, E.EngineModule):
    """Per-frame 2-D convolution on a video tensor (reference resnet.py:94-101)."""

    def run(self, x, g: E.Geom, *, x2=None, residual=None, out_scale=1.0, upsample=False, out_f32=False, rowbias=None,
            out_hw=None, out=None, gn_groups=None, hilo=False, out_hilo=False):
        """gn_groups: the output is (probably) normalised next by a GroupNorm of that many groups — the epilogue then
        reduces its statistics partials where it can (ops.conv_gemm); a wrong guess only costs the unused partials.
        hilo: x carries [hi | lo] fp16 halves of fp32 rows (2*C_in channels per pixel), the weights are repeated along C_in."""
        cw = E.packed_conv_dup(self, "w", self) if hilo else E.packed_conv(self, "w", self)
        return ops.conv_gemm(x, cw, a2=x2, n_img=g.n_img, t_len=g.t, hi=g.h, wi=g.w, stride=self.stride[0],
                             pad=(0, self.padding[0], self.padding[1]), upsample=upsample, residual=residual,
                             out_scale=out_scale, out_f32=out_f32, rowbias=rowbias, rows_per_batch=g.rows_per_batch,
                             out_hw=out_hw, out=out, gn_groups=gn_groups, out_hilo=out_hilo)

    def out_geom(self, g: E.Geom, upsample=False):
        if upsample:
            return g.with_hw(2 * g.h, 2 * g.w)
        s, p, k = self.stride[0], self.padding, self.kernel_size
        return g.with_hw((g.h + 2 * p[0] - k[0]) // s + 1, (g.w + 2 * p[1] - k[1]) // s + 1)

    def forward(self, x):
        rows, g = E.to_rows(x, c_pad=(self.in_channels if self.in_channels % 64 == 0 else 8))
        y = self.run(rows, g)
        return E.from_rows(y, self.out_geom(g), self.out_channels, out_dtype=x.dtype)


class Conv3dK11(nn.Conv3d, E.EngineModule):
    """nn.Conv3d container (temporal (k,1,1) and 3x3x3 kernels) executed by the implicit GEMM."""

    def run(self, x, g: E.Geom, *, residual=None, out_scale=1.0, rowbias=None, out_f32=False, gn_groups=None, out_hilo=False):
        cw = E.packed_conv(self, "w", self)
        return ops.conv_gemm(x, cw, n_img=g.n_img, t_len=g.t, hi=g.h, wi=g.w, stride=1, pad=tuple(self.padding),
                             residual=residual, out_scale=out_scale, rowbias=rowbias, rows_per_batch=g.rows_per_batch,
                             out_f32=out_f32, gn_groups=gn_groups, out_hilo=out_hilo)


class Upsample3D(E.EngineModule):
    """Nearest x(1,2,2) + 3x3 conv (reference resnet.py:104-158); the upsampling is folded into the
    convolution's gather, the 4x larger intermediate is never written."""

    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 interpolate_mode="nearest"):
        super().__init__()
        if use_conv_transpose or not use_conv or interpolate_mode != "nearest":
            raise NotImplementedError("only nearest + conv upsampling is on the hot path")
        self.channels, self.out_channels, self.name = channels, out_channels or channels, name
        conv = InflatedConv3d(self.channels, self.out_channels, 3, padding=1)
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv

    def run(self, x, g: E.Geom, output_size=None):
        conv = self.conv if self.name == "conv" else self.Conv2d_0
        s32 = x.dtype == torch.float32           # fp32 residual stream (VAE decoder): x is also this conv's operand
        hilo = s32 and E.SAMPLER_HILO in (True, "up") and self.channels % 64 == 0 and ("up", g.h) not in E.SAMPLER_HILO_SKIP
        x = E.hilo_rows(x) if hilo else ops.cast_f16(x)
        if output_size is None or tuple(output_size[-2:]) == (2 * g.h, 2 * g.w):
            g2 = g.with_hw(2 * g.h, 2 * g.w)
            if not PHASE_UPSAMPLE or tuple(conv.kernel_size) != (3, 3) or tuple(conv.padding) != (1, 1):
                return conv.run(x, g, upsample=True, out_f32=s32, gn_groups=E.GN_GROUPS_HINT, hilo=hilo), g2
            # four 2x2 convs on the low-resolution rows, each writing its sub-pixel phase of the output in place
            cws = E.packed_upsample_phases(conv, "w", conv, dup=hilo)
            out = torch.empty((g2.rows, self.out_channels), dtype=torch.float32 if s32 else torch.float16, device=x.device)
            # GroupNorm statistics of the output (its consumers: the next block's norm1, a TemporalModule3D): the four phase
            # launches write their partials into ONE workspace in which every output frame owns a contiguous run of chunks
            # (phase-major inside the frame) — the stand-alone statistics pass over the 4x larger tensor goes away
            sh, cpi = None, g.hw // 64
            if FUSE_UPSAMPLE_GN and ops.FUSE_GN_STATS and g.hw % 64 == 0 and self.out_channels % E.GN_GROUPS_HINT == 0:
                sh = ops.SharedGnPartials(g2.rows, E.GN_GROUPS_HINT, self.out_channels, x.device)
            for py in range(2):
                for px in range(2):
                    ops.conv_gemm(x, cws[py][px], n_img=g.n_img, t_len=g.t, hi=g.h, wi=g.w, pad=(0, 1 - py, 1 - px),
                                  out_hw=(g.h, g.w), out_f32=s32, rows_per_batch=g.rows_per_batch, out=out,
                                  out_map=(g.w, 4 * g.w, 2, py * 2 * g.w + px),
                                  gn_shared=None if sh is None else (sh, cpi, 4 * cpi, (2 * py + px) * cpi))
            if sh is not None and sh.filled == 4:
                ops._gn_attach(out, sh)
            return out, g2
        # Forced size (reference resnet.py:147-150, used when H, W are not multiples of 2^num_upsamplers,
        # unet_video.py:443-445,541-542): F.interpolate(size=..., mode="nearest"), i.e. src = floor(dst * in / out)
        # in fp32.  Rare path (odd intermediate sizes): the resized rows are materialised by an index gather and
        # the 3x3 conv then runs on the target geometry.
        ho, wo = int(output_size[-2]), int(output_size[-1])
        idx = self._nearest_rows(g, ho, wo, x.device)
        g2 = g.with_hw(ho, wo)
        return conv.run(x.index_select(0, idx), g2, out_f32=s32, gn_groups=E.GN_GROUPS_HINT, hilo=hilo), g2

    def _nearest_rows(self, g, ho, wo, device):
        key = (g.n_img, g.h, g.w, ho, wo, str(device))
        cache = self.__dict__.setdefault("_nearest_cache", {})
        if key not in cache:
            def src(n_in, n_out):
                scale = torch.tensor(n_in / n_out, dtype=torch.float32)
                return torch.clamp((torch.arange(n_out, dtype=torch.float32) * scale).floor().long(), max=n_in - 1)
            ys, xs = src(g.h, ho), src(g.w, wo)
            per_img = (ys[:, None] * g.w + xs[None, :]).reshape(-1)
            idx = (torch.arange(g.n_img)[:, None] * (g.h * g.w) + per_img[None, :]).reshape(-1)
            cache.clear()
            cache[key] = idx.to(device)
        return cache[key]

    def forward(self, hidden_states, output_size=None):
        rows, g = E.to_rows(hidden_states, c_pad=self.channels)
        y, g2 = self.run(rows, g, output_size)
        return E.from_rows(y, g2, self.out_channels, out_dtype=hidden_states.dtype)


class Downsample3D(E.EngineModule):
    """3x3 stride-2 conv (reference resnet.py:161-197)."""

    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        if not use_conv or padding not in (0, 1):
            raise NotImplementedError("only the stride-2 conv downsampler is on the hot path")
        self.channels, self.out_channels, self.padding, self.name = channels, out_channels or channels, padding, name
        conv = InflatedConv3d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        if name == "conv":
            self.Conv2d_0 = conv
        self.conv = conv

    def run(self, x, g: E.Geom):
        s32 = x.dtype == torch.float32           # fp32 residual stream: x is this conv's MFMA operand, the output is stream
        hilo = s32 and E.SAMPLER_HILO in (True, "down") and self.channels % 64 == 0 and ("down", g.h) not in E.SAMPLER_HILO_SKIP
        x = E.hilo_rows(x) if hilo else ops.cast_f16(x)
        if self.padding == 0:
            # reference pads (0,1,0,1) then convolves with pad 0 (resnet.py:188-192): the right /
            # bottom taps that fall outside read zeros in the gather, no padded copy is made
            g2 = g.with_hw((g.h + 1 - 3) // 2 + 1, (g.w + 1 - 3) // 2 + 1)
            return self.conv.run(x, g, out_hw=(g2.h, g2.w), out_f32=s32, gn_groups=E.GN_GROUPS_HINT, hilo=hilo), g2
        return self.conv.run(x, g, out_f32=s32, gn_groups=E.GN_GROUPS_HINT, hilo=hilo), self.conv.out_geom(g)

    def forward(self, hidden_states):
        rows, g = E.to_rows(hidden_states, c_pad=self.channels)
        y, g2 = self.run(rows, g)
        return E.from_rows(y, g2, self.out_channels, out_dtype=hidden_states.dtype)


class _ResnetBase(E.EngineModule):
    """Shared skeleton: GN -> SiLU -> conv1 (+temb) -> GN -> SiLU -> conv2 (+shortcut) / scale."""

    def _build(self, in_channels, out_channels, temb_channels, groups, groups_out, eps, output_scale_factor,
               time_embedding_norm, use_in_shortcut, make_conv, make_shortcut):
        if time_embedding_norm != "default":
            raise NotImplementedError("time_embedding_norm='scale_shift' is not used by the released configs")
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.output_scale_factor = output_scale_factor
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = make_conv(in_channels, out_channels, True)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = make_conv(out_channels, out_channels, False)
        self.use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = make_shortcut(in_channels, out_channels) if self.use_in_shortcut else None

    def run(self, x, g: E.Geom, temb=None, x2=None, c_real=None, stream_f32=None, out_f32=None, out_hilo=False):
        """x (and optional channel-concatenated x2): rows of in_channels; temb: fp32 [B][temb_ch].
        out_hilo (fp32 stream, identity shortcut): the block's fp32 output leaves as the fp16 hi | lo operand pair of a 1x1 consumer
        ([M][2 C], engine.tail_hilo), written by conv2's epilogue.
        out_f32=False: the block's OUTPUT is only ever read as an MFMA operand (TemporalModule3D's tail block feeds the
        1x1 shift_conv), so it is written in fp16 even when the block runs on an fp32 stream.

        Stream dtype: with fp32 rows in (or `stream_f32=True`) the block keeps its conv outputs, the residual sum and
        the GroupNorm inputs in fp32 — only the MFMA operands (GroupNorm outputs) are fp16, i.e. ONE fp16 rounding per
        conv instead of three (conv out, residual sum, norm out).  The VAE decoder runs this way by default because
        the reference decodes in fp32 (pipeline_upscale_a_video.py:668-681); the UNet follows its `stream_dtype`
        (unet_video.py)."""
        s32 = (x.dtype == torch.float32) if stream_f32 is None else bool(stream_f32)
        o32 = s32 if out_f32 is None else bool(out_f32)
        # the shortcut conv reads the (concatenated) input as an MFMA operand: with an fp32 stream its fp16 rounding is
        # written by the norm1 pass, which has the values in registers anyway
        raw16 = None
        want_raw = self.conv_shortcut is not None and x.dtype == torch.float32
        if want_raw and E.SHORTCUT_HILO and tuple(self.conv_shortcut.kernel_size) in ((1, 1), (1, 1, 1)) and c_real is None:
            want_raw = "hilo"
        h = E.group_norm(self, "norm1", self.norm1, x, n_inst=g.b, rows_per_inst=g.rows_per_batch, silu=True, x2=x2,
                         c_real=c_real, want_raw=want_raw)
        if want_raw:
            h, raw16 = h
        rb = _temb_rows(self, temb) if (temb is not None and self.time_emb_proj is not None) else None
        h = self.conv1.run(h, g, rowbias=rb, out_f32=s32 and E.branch_f32(), gn_groups=self.norm2.num_groups)
        h = E.group_norm(self, "norm2", self.norm2, h, n_inst=g.b, rows_per_inst=g.rows_per_batch, silu=True)
        osc = 1.0 / self.output_scale_factor
        if self.conv_shortcut is not None and raw16 is not None and E.FUSE_SHORTCUT and isinstance(self.conv2, InflatedConv3d) \
                and tuple(self.conv2.stride) == (1, 1) and tuple(self.conv_shortcut.kernel_size) == (1, 1) \
                and raw16.shape[-1] % 64 == 0 and self.conv2.in_channels % 64 == 0 and h.shape[-1] == self.conv2.in_channels:
            # (the packed shortcut weights sit at K offset conv2.in_channels: only when the branch rows carry exactly that many
            # channels — no channel padding — do the two sources line up with the packing; otherwise the unfused path below)
            # conv_shortcut([x | x2]) + conv2(h) in ONE implicit GEMM: the raw copy (fp16, or hi | lo halves) is the second
            # source and multiplies the centre tap only; no fp32 shortcut tensor is written and read back as the residual
            cw = E.packed_conv_with_shortcut(self, "conv2+shortcut", self.conv2, self.conv_shortcut, 2 if want_raw == "hilo" else 1)
            return ops.conv_gemm(h, cw, a2=raw16, a2_center=True, n_img=g.n_img, t_len=g.t, hi=g.h, wi=g.w,
                                 pad=(0, self.conv2.padding[0], self.conv2.padding[1]), out_scale=osc, out_f32=o32,
                                 rows_per_batch=g.rows_per_batch, gn_groups=self.norm2.num_groups)
        if self.conv_shortcut is not None:
            if raw16 is not None and want_raw == "hilo":      # [x | x2] as hi and lo fp16 halves: K = 2 * C_in, weights repeated
                res = ops.conv_gemm(raw16, E.packed_conv_hilo(self, "shortcut_hilo", self.conv_shortcut), n_img=g.n_img, t_len=g.t,
                                    hi=g.h, wi=g.w, out_f32=s32, rows_per_batch=g.rows_per_batch)
            elif raw16 is not None:               # [x | x2] already concatenated and rounded
                res = self.conv_shortcut.run(raw16, g, out_f32=s32)
            else:
                res = self.conv_shortcut.run(x, g, x2=x2, out_f32=s32) if x2 is not None else \
                    self.conv_shortcut.run(x, g, out_f32=s32)
        else:
            if x2 is not None:
                raise ops._lib.UavError("concatenated input needs a shortcut conv")
            if s32 and x.dtype != torch.float32:
                raise ops._lib.UavError("fp32 stream requested for an fp16 identity shortcut")
            res = x
        if out_hilo:
            if not o32:
                raise ops._lib.UavError("out_hilo needs an fp32 result")
            return self.conv2.run(h, g, residual=res, out_scale=osc, out_f32=True, out_hilo=True)
        # the block's output is the stream the next block normalises (with this block's group count, as a rule)
        return self.conv2.run(h, g, residual=res, out_scale=osc, out_f32=o32, gn_groups=self.norm2.num_groups)

    def forward(self, input_tensor, temb=None):
        cin = self.in_channels
        rows, g = E.to_rows(input_tensor, c_pad=(cin if cin % 64 == 0 else 8))
        t = None if temb is None else temb.float().contiguous()
        y = self.run(rows, g, t, c_real=(cin if cin % 64 else None))
        return E.from_rows(y, g, self.out_channels, out_dtype=input_tensor.dtype)


class ResnetBlock3D(_ResnetBase):
    """Per-frame 3x3 ResNet block with 5-D GroupNorm (reference resnet.py:200-294)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32,
                 groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", time_embedding_norm="default",
                 output_scale_factor=1.0, use_in_shortcut=None):
        super().__init__()
        self._build(in_channels, out_channels, temb_channels, groups, groups_out, eps, output_scale_factor,
                    time_embedding_norm, use_in_shortcut,
                    lambda i, o, first: InflatedConv3d(i, o, kernel_size=3, stride=1, padding=1),
                    lambda i, o: InflatedConv3d(i, o, kernel_size=1, stride=1, padding=0))


class ResnetBlock3DCNN(_ResnetBase):
    """Temporal ResNet block: Conv3d (k,1,1) then (3,1,1) (reference resnet.py:297-393)."""

    def __init__(self, *, in_channels, out_channels=None, kernel=(3, 1, 1), conv_shortcut=False, dropout=0.0,
                 temb_channels=512, groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish",
                 time_embedding_norm="default", output_scale_factor=1.0, use_in_shortcut=None):
        super().__init__()
        pad = tuple((kernel[i] - 1) // 2 for i in range(3))
        self._build(in_channels, out_channels, temb_channels, groups, groups_out, eps, output_scale_factor,
                    time_embedding_norm, use_in_shortcut,
                    lambda i, o, first: (Conv3dK11(i, o, kernel_size=kernel, stride=(1, 1, 1), padding=pad) if first
                                         else Conv3dK11(i, o, kernel_size=(3, 1, 1), stride=(1, 1, 1), padding=(1, 0, 0))),
                    lambda i, o: Conv3dK11(i, o, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0)))


class ResnetBlock3D_plus(ResnetBlock3D):
    """ResnetBlock3D + GroupNorm/SiLU/Conv3d(3,3,3) residual branch (reference resnet.py:396-500)."""

    def __init__(self, *, in_channels, out_channels=None, groups=32, groups_out=None, eps=1e-6, output_scale_factor=1.0,
                 **kw):
        super().__init__(in_channels=in_channels, out_channels=out_channels, groups=groups, groups_out=groups_out,
                         eps=eps, output_scale_factor=output_scale_factor, **kw)
        go = groups if groups_out is None else groups_out
        self.norm_3d = nn.GroupNorm(num_groups=go, num_channels=self.out_channels, eps=eps, affine=True)
        self.conv_3d = Conv3dK11(self.out_channels, self.out_channels, kernel_size=(3, 3, 3), stride=(1, 1, 1),
                                 padding=(1, 1, 1))

    def run(self, x, g: E.Geom, temb=None, x2=None, c_real=None, stream_f32=None, out_f32=None):
        out = super().run(x, g, temb, x2=x2, c_real=c_real, stream_f32=stream_f32)
        h = E.group_norm(self, "norm_3d", self.norm_3d, out, n_inst=g.b, rows_per_inst=g.rows_per_batch, silu=True)
        return self.conv_3d.run(h, g, residual=out, out_scale=1.0 / self.output_scale_factor,
                                out_f32=out.dtype == torch.float32, gn_groups=self.norm_3d.num_groups)


class Fuse_sft_block(E.EngineModule):
    """SFT conditioning of the decoder by the LR frames (reference resnet.py:63-79):
    out = dec + w*(dec*scale(e) + shift(e)), e = shared(cat[enc, dec])."""

    def __init__(self, enc_ch, dec_ch):
        super().__init__()
        self.shared = nn.Sequential(
            ResnetBlock3D(in_channels=enc_ch + dec_ch, out_channels=dec_ch, temb_channels=None),
            ResnetBlock3D(in_channels=dec_ch, out_channels=dec_ch, temb_channels=None))
        self.scale = InflatedConv3d(dec_ch, dec_ch, 3, 1, 1)
        self.shift = InflatedConv3d(dec_ch, dec_ch, 3, 1, 1)

    def run(self, enc, dec, g: E.Geom, w=1.0):
        s32 = dec.dtype == torch.float32
        e = self.shared[0].run(enc, g, None, x2=dec)
        e = ops.cast_f16(self.shared[1].run(e, g, None))
        scale = self.scale.run(e, g, out_f32=s32)
        shift = self.shift.run(e, g, out_f32=s32)
        return ops.sft_fuse(dec, scale, shift, w, out_f32=s32)          # dec + w*(dec*scale + shift)
