"""MI355X-native AutoencoderKLVideo (drop-in for the reference's
`models_video/autoencoder_kl_cond_video.py`: class :26-139, encode :173-185, _decode_cond :199-206,
decode :208-226).  Same config JSON (`vae_3d_config.json` / `vae_video_config.json`), same
state-dict keys; `decode(z, img, w_lr).sample` returns (B,3,T,4H,4W) fp32 like the reference.
Tiled / sliced decoding of the reference are never enabled by the CLI and are not built.
"""
from dataclasses import dataclass
from typing import Tuple

import torch

from uav import engine as E

from ._compat import BaseOutput, ConfigMixin, ModelMixin, register_to_config
from .resnet import InflatedConv3d
from .vae_video import Decoder, DecoderOutput, DiagonalGaussianDistribution, Encoder


@dataclass
class AutoencoderKLOutput(BaseOutput):
    latent_dist: "DiagonalGaussianDistribution"


class AutoencoderKLVideo(ModelMixin, ConfigMixin, E.EngineModule):
    _supports_gradient_checkpointing = False

    @register_to_config
    def __init__(self, in_channels: int = 3, out_channels: int = 3, down_block_types: Tuple[str] = ("DownEncoderBlock3D",),
                 up_block_types: Tuple[str] = ("UpDecoderBlock3D",), block_out_channels: Tuple[int] = (64,),
                 layers_per_block: int = 1, act_fn: str = "silu", latent_channels: int = 4, norm_num_groups: int = 32,
                 sample_size: int = 32, scaling_factor: float = 0.18215, condition_img: bool = False,
                 condition_channels: int = 128, use_temporal_block: bool = False):
        super().__init__()
        self.encoder = Encoder(in_channels=in_channels, out_channels=latent_channels, down_block_types=down_block_types,
                               block_out_channels=block_out_channels, layers_per_block=layers_per_block, act_fn=act_fn,
                               norm_num_groups=norm_num_groups, double_z=True)
        self.decoder = Decoder(in_channels=latent_channels, out_channels=out_channels, up_block_types=up_block_types,
                               block_out_channels=block_out_channels, layers_per_block=layers_per_block,
                               norm_num_groups=norm_num_groups, act_fn=act_fn, condition_img=condition_img,
                               condition_channels=condition_channels, use_temporal_block=use_temporal_block)
        self.quant_conv = InflatedConv3d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = InflatedConv3d(latent_channels, latent_channels, 1)
        self.use_slicing = False
        self.use_tiling = False
        # Decoder stream precision.  None (default): follow the parameter dtype like the reference does — an fp32 VAE (what
        # the CLI builds, inference_upscale_a_video.py:104-111, decoded in fp32 at pipeline:668-681) keeps conv outputs,
        # the residual stream and GroupNorm inputs in fp32 with fp16 MFMA operands (10-bit mantissa operands + fp32
        # accumulation: the arithmetic cuDNN's default TF32 convolutions give the reference on its own GPUs); `.half()`
        # selects the all-fp16 rows of round 1 (faster, ~2x the error).  torch.float32 / torch.float16 force a mode.
        self.stream_dtype = None

    @E.guarded
    def encode(self, x, return_dict: bool = True):
        rows, g = E.to_rows(x, c_pad=8)
        h, g2 = self.encoder.run(rows, g)                                     # fp32 rows [..][2*latent]
        moments = self.quant_conv.run(h.half().contiguous(), g2, out_f32=True)
        m = E.from_rows(moments, g2, 2 * self.config.latent_channels, out_dtype=torch.float32)
        post = DiagonalGaussianDistribution(m)
        return AutoencoderKLOutput(latent_dist=post) if return_dict else (post,)

    @E.guarded
    def decode_rows(self, z, img=None, w_lr=1.0, latent_scale=1.0):
        """z (B,4,T,H,W) -> fp32 output rows [B*T*4H*4W][4] (3 real channels) + geometry."""
        zr, g = E.to_rows(z, c_pad=8, scale=latent_scale)
        # 1x1 conv 4 -> 4 channels, written into 8-channel rows (the padding columns stay zero)
        zq = torch.zeros((zr.shape[0], 8), dtype=torch.float16, device=zr.device)
        self.post_quant_conv.run(zr, g, out=zq)
        ir = None
        if img is not None and self.decoder.condition_img:
            ir, _ = E.to_rows(img, c_pad=8)
        sd = self.stream_dtype if self.stream_dtype is not None else self.decoder.conv_in.weight.dtype
        return self.decoder.run(zq, g, ir, w_lr, stream_f32=sd == torch.float32)

    def decode(self, z, img=None, w_lr=1, return_dict: bool = True, clamp=None):
        y, g2 = self.decode_rows(z, img, float(w_lr))
        dec = E.from_rows(y, g2, self.config.out_channels, out_dtype=torch.float32, clamp=clamp)
        return DecoderOutput(sample=dec) if return_dict else (dec,)

    def forward(self, sample, sample_posterior: bool = False, return_dict: bool = True, generator=None):
        post = self.encode(sample).latent_dist
        z = post.sample(generator=generator) if sample_posterior else post.mode()
        return self.decode(z, return_dict=return_dict)
