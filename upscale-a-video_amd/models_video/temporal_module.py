"""MI355X-native TemporalModule3D (drop-in for the reference's `models_video/temporal_module.py`
:98-194; the transformer / DCN variants in that file are disabled by the released config,
`attention_block_types=["",""]`, and are not part of the hot path).

forward: ResnetBlock3DCNN (5,1,1)+(3,1,1) with temb -> ResnetBlock3D 3x3+3x3 with temb ->
1x1 `shift_conv` -> input + h; the final add rides in the shift_conv GEMM epilogue.
"""
import torch
import torch.nn as nn

from uav import engine as E

from .resnet import InflatedConv3d, ResnetBlock3D, ResnetBlock3DCNN


class EmptyTemporalModule3D(nn.Module):
    def run(self, x, g, temb=None):
        return x

    def forward(self, hidden_states, w=1.0, encoder_hidden_states=None, timesteps=None, temb=None, attention_mask=None):
        return hidden_states


class TemporalModule3D(E.EngineModule):
    def __init__(self, in_channels=None, out_channels=None, num_attention_layers=1, num_attention_head=8,
                 attention_head_dim=None, cross_attention_dim=768, temb_channels=512, dropout=0.0, attention_bias=False,
                 activation_fn="geglu", only_cross_attention=False, upcast_attention=False, norm_num_groups=8,
                 use_linear_projection=True, use_scale_shift=False, attention_block_types=("", ""),
                 cross_frame_attention_mode=None, temporal_shift_fold_div=None, temporal_shift_direction="right",
                 use_dcn_warpping=False, use_deformable_conv=True, attention_dim_div=2):
        super().__init__()
        if tuple(attention_block_types) != ("", "") or use_scale_shift:
            raise NotImplementedError("temporal transformer / scale-shift variants are disabled in the released config")
        self.in_channels = in_channels
        self.resblocks_3d_temporal = ResnetBlock3DCNN(in_channels=in_channels, out_channels=in_channels, kernel=(5, 1, 1),
                                                      temb_channels=temb_channels)
        self.resblocks_3d_spatial = ResnetBlock3D(in_channels=in_channels, out_channels=in_channels,
                                                  temb_channels=temb_channels, groups=32, groups_out=32)
        self.shift_conv = InflatedConv3d(in_channels=in_channels, out_channels=in_channels, kernel_size=1, stride=1, padding=0)
        for p in self.shift_conv.parameters():        # zero-initialised in the reference (temporal_module.py:172)
            p.detach().zero_()

    def run(self, x, g: E.Geom, temb=None, w=1.0):
        s32 = x.dtype == torch.float32           # fp32 residual stream (UNetVideoModel.stream_dtype)
        h = self.resblocks_3d_temporal.run(x, g, temb)
        tail_hilo = s32 and E.tail_hilo() and self.in_channels % 64 == 0
        # the tail block's output is only read as shift_conv's MFMA operand: fp16, or (TAIL_HILO) fp32 rows as a hi | lo pair
        fused_pair = tail_hilo and self.resblocks_3d_spatial.conv_shortcut is None      # conv2's epilogue writes the pair (no cast pass)
        h = self.resblocks_3d_spatial.run(h, g, temb, out_f32=None if tail_hilo else False, out_hilo=fused_pair)
        if w != 1.0:
            raise NotImplementedError("w != 1 is never used by the pipeline")
        if tail_hilo:
            return self.shift_conv.run(h if fused_pair else E.hilo_rows(h), g, residual=x, out_f32=s32, gn_groups=E.GN_GROUPS_HINT, hilo=True)
        return self.shift_conv.run(h, g, residual=x, out_f32=s32, gn_groups=E.GN_GROUPS_HINT)

    def forward(self, hidden_states, w=1, encoder_hidden_states=None, timesteps=None, temb=None, attention_mask=None):
        rows, g = E.to_rows(hidden_states, c_pad=self.in_channels)
        t = None if temb is None else temb.float().contiguous()
        return E.from_rows(self.run(rows, g, t, w), g, self.in_channels, out_dtype=hidden_states.dtype)
