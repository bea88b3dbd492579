"""MI355X-native video-VAE encoder/decoder stacks (drop-in for the reference's
`models_video/vae_video.py`: Encoder :55, Decoder :264, DiagonalGaussianDistribution :407).

Only the DECODER is on the inference hot path (the pipeline never encodes, SURVEY.md headline
facts); the Encoder is built from the same HIP-backed blocks so that checkpoints load with
strict=True and `encode` works, but it is not tuned.  Decoder.forward (reference :365-405):
conv_in 4->C -> [video VAE: SFT conditioning on the LR frames] -> mid (ResNet, single-head
attention over H*W tokens, ResNet) -> 3 up blocks (3 ResNets, nearest-2x + conv) ->
GroupNorm -> SiLU -> conv_out, all channels-last fp16 with fp32 accumulation; the reference runs
this stage in fp32 (pipeline_upscale_a_video.py:668) — BASELINE.json's north star asks for the
MFMA path, and the fp16-vs-fp32 difference is reported in DESIGN.md.
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from uav import engine as E

from ._compat import BaseOutput
from .resnet import Downsample3D, Fuse_sft_block, InflatedConv3d, ResnetBlock3D, ResnetBlock3D_plus
from .temporal_module import EmptyTemporalModule3D  # noqa: F401  (name kept for import parity)
from .unet_blocks import UNetMidBlock3D, UNetMidBlock3D_plus, get_up_block


@dataclass
class DecoderOutput(BaseOutput):
    sample: torch.FloatTensor


class DownEncoderBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_downsample=True,
                 downsample_padding=1):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=None, eps=resnet_eps,
                                                    groups=resnet_groups) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None

    def run(self, x, g):
        for r in self.resnets:
            x = r.run(x, g, None)
        if self.downsamplers is not None:
            x, g = self.downsamplers[0].run(x, g)
        return x, g


class Encoder(E.EngineModule):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock3D",), block_out_channels=(64,),
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", double_z=True):
        super().__init__()
        self.layers_per_block = layers_per_block
        self.in_channels = in_channels
        self.conv_in = InflatedConv3d(in_channels, block_out_channels[0], kernel_size=3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        for i, _ in enumerate(down_block_types):
            input_channel, output_channel = output_channel, block_out_channels[i]
            self.down_blocks.append(DownEncoderBlock3D(input_channel, output_channel, num_layers=layers_per_block,
                                                       resnet_eps=1e-6, resnet_groups=norm_num_groups,
                                                       add_downsample=i != len(block_out_channels) - 1, downsample_padding=0))
        self.mid_block = UNetMidBlock3D(in_channels=block_out_channels[-1], resnet_eps=1e-6, resnet_act_fn=act_fn,
                                        output_scale_factor=1, resnet_time_scale_shift="default",
                                        attn_num_head_channels=None, resnet_groups=norm_num_groups, temb_channels=None)
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[-1], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(block_out_channels[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)
        self.gradient_checkpointing = False

    def run(self, x, g):
        x = self.conv_in.run(x, g)
        for blk in self.down_blocks:
            x, g = blk.run(x, g)
        x = self.mid_block.run(x, g)
        x = E.group_norm(self, "conv_norm_out", self.conv_norm_out, x, n_inst=g.b, rows_per_inst=g.rows_per_batch, silu=True)
        return self.conv_out.run(x, g, out_f32=True), g

    def forward(self, x):
        rows, g = E.to_rows(x, c_pad=8)
        y, g2 = self.run(rows, g)
        return E.from_rows(y, g2, self.conv_out.out_channels, out_dtype=torch.float32)


class Decoder(E.EngineModule):
    def __init__(self, in_channels=3, out_channels=3, up_block_types=("UpDecoderBlock3D",), block_out_channels=(64,),
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", condition_img=False, condition_channels=128,
                 use_temporal_block=False):
        super().__init__()
        self.layers_per_block = layers_per_block
        self.use_temporal_block = use_temporal_block
        self.condition_img = condition_img
        self.out_channels = out_channels
        self.mid_block_type = "UNetMidBlock3D" if up_block_types[0] == "UpDecoderBlock3D" else "UNetMidBlock3D_plus"
        self.conv_in = InflatedConv3d(in_channels, block_out_channels[-1], kernel_size=3, stride=1, padding=1)
        if self.condition_img:
            self.condition_in = nn.Sequential(
                ResnetBlock3D_plus(in_channels=3, out_channels=condition_channels, temb_channels=None, groups=3, groups_out=32),
                ResnetBlock3D_plus(in_channels=condition_channels, out_channels=condition_channels, temb_channels=None))
            self.condition_fuse = Fuse_sft_block(condition_channels, block_out_channels[-1])
        mid_cls = UNetMidBlock3D if self.mid_block_type == "UNetMidBlock3D" else UNetMidBlock3D_plus
        self.mid_block = mid_cls(in_channels=block_out_channels[-1], resnet_eps=1e-6, resnet_act_fn=act_fn,
                                 output_scale_factor=1, resnet_time_scale_shift="default", attn_num_head_channels=None,
                                 resnet_groups=norm_num_groups, temb_channels=None)
        self.up_blocks = nn.ModuleList([])
        if self.use_temporal_block:
            self.mid_temporal_block = None
            self.up_temporal_blocks = nn.ModuleList([])
        rev = list(reversed(block_out_channels))
        output_channel = rev[0]
        for i, up_block_type in enumerate(up_block_types):
            prev_output_channel, output_channel = output_channel, rev[i]
            self.up_blocks.append(get_up_block(up_block_type, num_layers=layers_per_block + 1, in_channels=prev_output_channel,
                                               out_channels=output_channel, prev_output_channel=None,
                                               add_upsample=i != len(block_out_channels) - 1, resnet_eps=1e-6,
                                               resnet_act_fn=act_fn, resnet_groups=norm_num_groups,
                                               attn_num_head_channels=None, temb_channels=None))
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(block_out_channels[0], out_channels, 3, padding=1)
        self.gradient_checkpointing = False

    def run(self, z, g, img=None, w_lr=1.0, stream_f32=False):
        """z: latent rows [..][8] (after post_quant_conv); img: LR frame rows [..][8] (3 real channels).
        stream_f32: conv outputs / residual stream / GroupNorm inputs in fp32 (see _ResnetBase.run)."""
        x = self.conv_in.run(z, g, out_f32=stream_f32, gn_groups=E.GN_GROUPS_HINT)
        if self.condition_img:
            if img is None:
                raise AssertionError("input img condition when condition_img is True.")
            c = self.condition_in[0].run(img, g, None, c_real=3, stream_f32=stream_f32)
            c = self.condition_in[1].run(c, g, None)
            x = self.condition_fuse.run(c, x, g, w=w_lr)
        x = self.mid_block.run(x, g)
        for blk in self.up_blocks:
            x, g = blk.run(x, g)
        x = E.group_norm(self, "conv_norm_out", self.conv_norm_out, x, n_inst=g.b, rows_per_inst=g.rows_per_batch, silu=True)
        return self.conv_out.run(x, g, out_f32=True), g


class DiagonalGaussianDistribution(object):
    """Posterior of the (off-hot-path) encoder; plain tensor math on the tiny latent."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None):
        noise = torch.randn(self.mean.shape, generator=generator, device="cpu" if generator is not None and generator.device.type == "cpu" else self.mean.device,
                            dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean
