"""MI355X-native UNetVideoModel (drop-in for the reference's `models_video/unet_video.py`:
class UNetVideoModel :103, forward :404-574).

Same constructor / config JSON (`configs/unet_video_config.json`), same 1158 state-dict keys and
the same `forward(sample, timestep, low_res, encoder_hidden_states, class_labels)` ->
`UNet3DConditionOutput(sample=...)` contract.  Inside, the video tensor is converted ONCE to
channels-last fp16 rows (B,T,H,W,C) and stays that way until `conv_out`; every op is a HIP kernel
of libuav_hip.so (see resnet.py / attention.py / temporal_module.py for the per-module mapping).
Differences that do not change results: no `torch.cat` for the 7-channel input or the skip
connections, the time/class embedding MLP runs in fp32, no host synchronisation inside forward
(the reference syncs on `torch.any(class_labels > max)`, unet_video.py:484 — the check is done on
the host copy when one exists).
"""
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import os

import torch
import torch.nn as nn

from uav import engine as E
from uav import ops

# CFG-shared head: its skip tensors are kept once and read batch-broadcast (UAV_BROADCAST_SKIPS=0: duplicated with cat)
BROADCAST_SKIPS = os.environ.get("UAV_BROADCAST_SKIPS", "1") != "0"
# Default residual-stream precision (DESIGN.md §4 has the table the choice rests on): fp32 rows meet the stated 1e-3 at the
# headline shape (9.8e-4 vs 1.68e-3 for fp16 rows) and over the whole 30-step schedule (1.0e-3 vs 2.2e-3; the reference's
# own half pipeline: 3.0e-3) for ~10 % of the clip time; "f16" is the reference's `.half()` arithmetic.
DEFAULT_STREAM = "f32"

from ._compat import BaseOutput, ConfigMixin, ModelMixin, register_to_config
from .attention import RotaryEmbedding
from .resnet import InflatedConv3d
from .temporal_module import EmptyTemporalModule3D, TemporalModule3D
from .unet_blocks import (CrossAttnDownBlock3D, CrossAttnUpBlock3D, DownBlock3D, UNetMidBlock3DCrossAttn, UpBlock3D,
                          get_down_block, get_up_block)


@dataclass
class UNet3DConditionOutput(BaseOutput):
    sample: torch.FloatTensor


class Timesteps(nn.Module):
    """Parameter-free sinusoidal embedding (diffusers 0.16.0 `Timesteps`)."""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift


class TimestepEmbedding(nn.Module):
    """linear_1 -> SiLU -> linear_2 (diffusers 0.16.0; parameter names matter for the state dict)."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class UNetVideoModel(ModelMixin, ConfigMixin, E.EngineModule):
    _supports_gradient_checkpointing = False

    @register_to_config
    def __init__(
        self,
        down_temporal_idx=(0, 1, 2),
        mid_temporal=False,
        up_temporal_idx=(1, 2, 3),
        temporal_module_config=None,
        sample_size: Optional[int] = None,
        in_channels: int = 7,
        out_channels: int = 4,
        center_input_sample: bool = False,
        max_noise_level: int = 350,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        attention_head_dim: Union[int, Tuple[int]] = 8,
        block_out_channels: Tuple[int] = (256, 512, 512, 1024),
        down_block_types: Tuple[str] = ("DownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D"),
        mid_block_type: str = "UNetMidBlock3DCrossAttn",
        up_block_types: Tuple[str] = ("CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "UpBlock3D"),
        only_cross_attention: Union[bool, Tuple[bool]] = (True, True, True, False),
        layers_per_block: int = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: int = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: int = 1024,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = True,
        class_embed_type: Optional[str] = None,
        num_class_embeds: Optional[int] = 1000,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        use_first_frame: bool = False,
        use_relative_position: bool = False,
    ):
        super().__init__()
        if in_channels > 8:
            raise NotImplementedError("conv_in expects <= 8 input channels (4 latent + 3 low-res)")
        if class_embed_type is not None:
            raise NotImplementedError("only the nn.Embedding noise-level class embedding is used by the release")
        temporal_module_config = temporal_module_config or {}
        self.sample_size = sample_size
        time_embed_dim = block_out_channels[0] * 4
        self.conv_in = InflatedConv3d(in_channels, block_out_channels[0], kernel_size=3, padding=1)
        self.time_proj = Timesteps(block_out_channels[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], time_embed_dim)
        self.class_embedding = nn.Embedding(num_class_embeds, time_embed_dim) if num_class_embeds is not None else None

        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        self.down_temp_blocks = nn.ModuleList([])
        self.up_temp_blocks = nn.ModuleList([])
        if isinstance(only_cross_attention, bool):
            only_cross_attention = [only_cross_attention] * len(down_block_types)
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * len(down_block_types)
        self.temporal_rotary_emb = RotaryEmbedding(32)

        common = dict(resnet_eps=norm_eps, resnet_act_fn=act_fn, resnet_groups=norm_num_groups,
                      cross_attention_dim=cross_attention_dim, dual_cross_attention=dual_cross_attention,
                      use_linear_projection=use_linear_projection, upcast_attention=upcast_attention,
                      resnet_time_scale_shift=resnet_time_scale_shift, use_first_frame=use_first_frame,
                      use_relative_position=use_relative_position, rotary_emb=self.temporal_rotary_emb)
        output_channel = block_out_channels[0]
        for i, down_block_type in enumerate(down_block_types):
            input_channel, output_channel = output_channel, block_out_channels[i]
            is_final = i == len(block_out_channels) - 1
            self.down_blocks.append(get_down_block(
                down_block_type, num_layers=layers_per_block, in_channels=input_channel, out_channels=output_channel,
                temb_channels=time_embed_dim, add_downsample=not is_final, attn_num_head_channels=attention_head_dim[i],
                downsample_padding=downsample_padding, only_cross_attention=only_cross_attention[i], **common))
            self.down_temp_blocks.append(
                TemporalModule3D(in_channels=output_channel, out_channels=output_channel, temb_channels=time_embed_dim,
                                 **temporal_module_config) if i in down_temporal_idx else EmptyTemporalModule3D())

        if mid_block_type != "UNetMidBlock3DCrossAttn":
            raise ValueError(f"unknown mid_block_type : {mid_block_type}")
        self.mid_block = UNetMidBlock3DCrossAttn(
            in_channels=block_out_channels[-1], temb_channels=time_embed_dim, resnet_eps=norm_eps, resnet_act_fn=act_fn,
            output_scale_factor=mid_block_scale_factor, resnet_time_scale_shift=resnet_time_scale_shift,
            cross_attention_dim=cross_attention_dim, attn_num_head_channels=attention_head_dim[-1],
            resnet_groups=norm_num_groups, dual_cross_attention=dual_cross_attention,
            use_linear_projection=use_linear_projection, upcast_attention=upcast_attention,
            use_first_frame=use_first_frame, use_relative_position=use_relative_position,
            rotary_emb=self.temporal_rotary_emb)
        self.mid_temp_block = (TemporalModule3D(in_channels=block_out_channels[-1], out_channels=block_out_channels[-1],
                                                temb_channels=time_embed_dim, **temporal_module_config)
                               if mid_temporal else EmptyTemporalModule3D())

        self.num_upsamplers = 0
        rev_ch = list(reversed(block_out_channels))
        rev_hd = list(reversed(attention_head_dim))
        rev_oca = list(reversed(only_cross_attention))
        output_channel = rev_ch[0]
        for i, up_block_type in enumerate(up_block_types):
            is_final = i == len(block_out_channels) - 1
            prev_output_channel, output_channel = output_channel, rev_ch[i]
            input_channel = rev_ch[min(i + 1, len(block_out_channels) - 1)]
            if not is_final:
                self.num_upsamplers += 1
            self.up_blocks.append(get_up_block(
                up_block_type, num_layers=layers_per_block + 1, in_channels=input_channel, out_channels=output_channel,
                prev_output_channel=prev_output_channel, temb_channels=time_embed_dim, add_upsample=not is_final,
                attn_num_head_channels=rev_hd[i], only_cross_attention=rev_oca[i], **common))
            self.up_temp_blocks.append(
                TemporalModule3D(in_channels=output_channel, out_channels=output_channel, temb_channels=time_embed_dim,
                                 **temporal_module_config) if i in up_temporal_idx else EmptyTemporalModule3D())

        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(block_out_channels[0], out_channels, kernel_size=3, padding=1)
        # Precision of the residual stream (conv outputs, residual sums, skip tensors, GroupNorm / LayerNorm INPUTS).
        # torch.float32: fp32 rows, only the MFMA operands (norm outputs, attention q/k/v/p, GEGLU output) are fp16 — one
        # fp16 rounding per contraction instead of one per stored tensor; torch.float16: every stored tensor is fp16, the
        # arithmetic of the reference's `.half()` UNet (inference_upscale_a_video.py:113-118).  None: UAV_UNET_STREAM
        # (f32 | f16), see DESIGN.md §4 for the measured parity / cost of both.
        self.stream_dtype = None
        self.__dict__["_env_stream"] = os.environ.get("UAV_UNET_STREAM", DEFAULT_STREAM)      # read once, at construction
        # precision = "high" (attribute, `UAV_UNET_PRECISION=high`, bench.py --precision high): the two remaining fp16 roundings of
        # the fp32 stream that are cheap to lift — block tails read by their 1x1 consumer as an fp16 hi | lo pair (engine.TAIL_HILO)
        # and the ResNet branch tensor between conv1 and norm2 kept in fp32 (engine.BRANCH_F32).  Measured at the headline shape
        # (round 4, profiles/r04_parity_precision_knobs_at_headline_shape_run3.jsonl): latents 8.2e-4 -> 7.1e-4, `.images` 9.5e-4 ->
        # 8.5e-4 over all pixels / 1.23e-3 -> 1.10e-3 over the pixels the reference does not clamp, for +5.7 % per clip.
        # "default": both off.  The switches live in uav.engine and are set for the duration of forward().
        self.precision = os.environ.get("UAV_UNET_PRECISION", "default")

    # ------------------------------------------------------------------------------------------
    def _embedding(self, timestep, class_labels, bsz, dev):
        """time_embedding(time_proj(t)) + class_embedding(noise_level) -> fp32 [B][4*C0]
        (reference unet_video.py:457-491)."""
        if torch.is_tensor(timestep):
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        else:
            t = torch.full((1,), float(timestep), dtype=torch.float32, device=dev)
        t = t.expand(bsz).contiguous()
        temb = ops.timestep_embedding(t, self.time_proj.num_channels, self.time_proj.flip_sin_to_cos,
                                      float(self.time_proj.downscale_freq_shift))
        te = self.time_embedding
        h = ops.linear_small(temb, E.f16_param(self, "te1.w", te.linear_1.weight), E.f32_param(self, "te1.b", te.linear_1.bias),
                             post_silu=True)
        emb = ops.linear_small(h, E.f16_param(self, "te2.w", te.linear_2.weight), E.f32_param(self, "te2.b", te.linear_2.bias))
        if self.class_embedding is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            if not torch.is_tensor(class_labels):
                class_labels = torch.tensor([int(class_labels)], dtype=torch.long)
            if class_labels.device.type == "cpu" and bool((class_labels > self.config.max_noise_level).any()):
                raise ValueError(f"`noise_level` has to be <= {self.config.max_noise_level} but is {class_labels}")
            tab = E.f32_param(self, "class_emb", self.class_embedding.weight)
            if class_labels.device.type == "cpu" and torch.device(dev).type != "cpu":
                # the pipeline hands the noise level as a host tensor on every call: a pageable host->device copy waits for the
                # stream to drain, i.e. one host/GPU rendezvous per UNet forward.  Keep the device copy per label value.
                key = (tuple(int(v) for v in class_labels.reshape(-1).tolist()), str(dev))
                cache = self.__dict__.setdefault("_class_label_dev", {})
                idx = cache.get(key)
                if idx is None:
                    if len(cache) >= 16:
                        cache.clear()
                    idx = E.publish(class_labels.to(dev).reshape(-1))
                    cache[key] = idx
                E.acquire(idx)
            else:
                idx = class_labels.to(dev).reshape(-1)
            ce = tab.index_select(0, idx)                                        # embedding lookup (gather)
            emb = emb + ce                                                       # broadcast (1|B, D)
        return emb.contiguous()

    def stream_f32(self):
        """Residual-stream precision of this model: `stream_dtype` when set, else UAV_UNET_STREAM as it stood when the model was
        BUILT (resolved once in __init__, not per forward), else DEFAULT_STREAM."""
        sd = self.stream_dtype
        if sd is None:
            return self.__dict__.get("_env_stream", DEFAULT_STREAM) == "f32"
        return sd == torch.float32

    @E.guarded
    def forward(self, sample, timestep, low_res, encoder_hidden_states=None, class_labels=20, attention_mask=None,
                return_dict: bool = True, cfg_shared_input: bool = False):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed by the pipeline")
        if sample.shape[1] + low_res.shape[1] != self.config.in_channels:
            raise ValueError(f"expected {self.config.in_channels} input channels, got {sample.shape[1]}+{low_res.shape[1]}")
        if self.precision not in ("default", "high"):
            raise ValueError(f"UNetVideoModel.precision must be 'default' or 'high', got {self.precision!r}")
        if self.precision == "high" and not (E.tail_hilo() and E.branch_f32()):
            with E.precision_high():         # thread-local scope: other host threads sharing this UNet are not affected
                return self.forward(sample, timestep, low_res, encoder_hidden_states, class_labels, attention_mask, return_dict,
                                    cfg_shared_input)
        dev = sample.device
        s32 = self.stream_f32()
        if self.config.center_input_sample:
            # the reference centres the CONCATENATED [sample, low_res] tensor (unet_video.py:440-454)
            sample = 2 * sample - 1.0
            low_res = 2 * low_res - 1.0
        x, g = E.to_rows(sample, c_pad=8, x5b=low_res)                           # cat on C (4+3 -> 8 padded)
        bsz = sample.shape[0]
        emb = self._embedding(timestep, class_labels, bsz, dev)

        ehs = encoder_hidden_states
        # fp16 rows of the prompt embeddings, kept per prompt TENSOR (identity + version) so that the blocks' text K/V caches,
        # keyed on these rows, hit from the second call on; a few entries because callers alternate between tensors (the
        # guidance branches evaluated one by one: pipeline.shard_cfg / overlap_streams; two clips with different prompts)
        srcs = self.__dict__.get("_ehs_src", ())
        src = next((s_ for s_ in srcs if s_[0] is ehs and s_[1] == ehs._version), None)
        if src is None:
            rows = E.publish(ehs.to(device=dev, dtype=torch.float16).reshape(-1, ehs.shape[-1]).contiguous())
            src = (ehs, ehs._version, rows)
            self.__dict__["_ehs_src"] = (src,) + tuple(srcs)[:3]
        ehs_rows = E.acquire(src[2])           # the local tuple, not the attribute: another stream's thread may have replaced it
        n_text = ehs.shape[1]

        # Classifier-free guidance feeds the same latents / low_res / timestep to both batch entries and only the
        # text differs, so everything ahead of the first text-conditioned block is computed once and duplicated
        # (bit-identical to running it twice; the pipeline sets `cfg_shared_input`, default off).
        first = 0
        if cfg_shared_input and bsz == 2 and not getattr(self.down_blocks[0], "has_cross_attention", True):
            rows1 = g.rows // 2
            g1 = E.Geom(1, g.t, g.h, g.w)
            emb1 = emb[:1].contiguous()
            x1 = self.conv_in.run(x[:rows1], g1, out_f32=s32)
            skips1 = [(x1, g1)]
            x1, g1, outs = self.down_blocks[0].run(x1, g1, emb1, ehs_rows, n_text)
            skips1.extend(outs)
            x1 = self.down_temp_blocks[0].run(x1, g1, emb1)
            # the skips stay single: their consumers (GroupNorm and convs with a second source) read them batch-broadcast
            skips = [((s_ if BROADCAST_SKIPS else torch.cat([s_, s_])), E.Geom(2, sg.t, sg.h, sg.w)) for (s_, sg) in skips1]
            x, g = ops.duplicate_rows(x1), E.Geom(2, g1.t, g1.h, g1.w)
            first = 1
        else:
            x = self.conv_in.run(x, g, out_f32=s32)
            skips = [(x, g)]
        for blk, tblk in list(zip(self.down_blocks, self.down_temp_blocks))[first:]:
            x, g, outs = blk.run(x, g, emb, ehs_rows, n_text)
            skips.extend(outs)
            x = tblk.run(x, g, emb)
        x = self.mid_block.run(x, g, emb, ehs_rows, n_text)
        x = self.mid_temp_block.run(x, g, emb)
        for blk, tblk in zip(self.up_blocks, self.up_temp_blocks):
            x, g = blk.run(x, g, skips, emb, ehs_rows, n_text)
            x = tblk.run(x, g, emb)
        x = E.group_norm(self, "conv_norm_out", self.conv_norm_out, x, n_inst=g.b, rows_per_inst=g.rows_per_batch, silu=True)
        y = self.conv_out.run(x, g, out_f32=True)
        out = E.from_rows(y, g, self.config.out_channels, out_dtype=sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float16)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop("_ehs_src", None)
        return super()._apply(fn, *a, **kw)
