"""MI355X-native flow-guided latent propagation (drop-in for the reference's
`models_video/propagation_module.py`: flow_warp :104-135, fbConsistencyCheck :140-149,
Propagation.forward :194-281).  Inference uses the parameter-free branch (`learnable=False`,
inference_upscale_a_video.py:126); the learnable DCN branch (:284-371) is training-only and is
not built.

Each recurrence step (consistency mask from the bilinearly warped check-flow, nearest/bilinear
warp of the propagated frame, fuse, mask blend) is ONE kernel launch `uav_propagate_step_f16`
working in place on frame planes of the (1,C,T,H,W) tensor: 2*(T-1) launches per call instead of
~25 ATen ops per step.  For fp16 latents the grid arithmetic is replayed in fp16 exactly as the
reference's GPU path evaluates it (grid built in the latent dtype, :123-132), because the
nearest-neighbour index is discontinuous in the coordinates; fp32 latents (the UNet on an fp32
residual stream) are warped unrounded on fp32 grids (`uav_propagate_step_f32`), which is what the
reference's fp32 run computes.
"""
import torch
import torch.nn as nn

from uav import engine as E
from uav import ops


class EmptyPropagation(nn.Module):
    def forward(self, feats_in, flows_forward, flows_backward):
        return feats_in


class Propagation(nn.Module):
    def __init__(self, in_channels, mid_channels=256, max_residue_magnitude=10, num_blocks=2, learnable=True):
        super().__init__()
        if learnable:
            raise NotImplementedError("the learnable (DCN) propagation branch is training-only; inference uses learnable=False")
        self.learnable = False
        self.module = ["backward_prop", "forward_prop"]
        self.coord_f16 = None          # None: follow the latent dtype like the reference; True/False to force

    @E.guarded
    def forward(self, x, flows_forward, flows_backward, interpolation="bilinear", mode="fuse", fuse_scale=0.5,
                alpha1=0.01, alpha2=0.5):
        b, c, t, h, w = x.shape
        if b != 1:
            raise NotImplementedError("the pipeline propagates one clip at a time (batch 1)")
        if flows_forward.shape[2] != t - 1 or flows_backward.shape[2] != t - 1:
            raise NotImplementedError("flows must carry T-1 frames (the reference's trilinear-in-time area resize is never "
                                      "exercised: RAFT_bi returns T-1 flows)")
        if tuple(flows_forward.shape[-2:]) != (h, w) or tuple(flows_backward.shape[-2:]) != (h, w):
            # reference :206-209: F.interpolate(flows, (t-1, h, w), mode='area') * (w / w_f) — identity for this pipeline
            # (latents live at LR resolution), needed when a caller hands over flows of another resolution
            s = 1.0 * w / flows_forward.shape[-1]
            flows_forward = ops.resize_area_f32(flows_forward.float(), h, w, mul=s).to(flows_forward.dtype)
            flows_backward = ops.resize_area_f32(flows_backward.float(), h, w, mul=s).to(flows_backward.dtype)
        if mode == "copy":
            fuse_scale = 1.0
        elif mode != "fuse":
            raise ValueError(mode)
        # The reference casts the flows to the latent dtype (pipeline:651) and builds the grid in it (:123): a half
        # pipeline warps fp16 values on fp16 grids, an fp32 one fp32 values on fp32 grids.  Same here: fp32 latents go
        # through `uav_propagate_step_f32` unrounded; `coord_f16 = True` on fp32 input forces the fp16 replay (values
        # rounded to fp16 as well — what round 3 did unconditionally).
        coord_f16 = (x.dtype == torch.float16) if self.coord_f16 is None else self.coord_f16
        work = torch.float32 if (x.dtype == torch.float32 and not coord_f16) else torch.float16
        xin = x.to(work).contiguous()
        ff = flows_forward.to(work).contiguous()
        fb = flows_backward.to(work).contiguous()
        nearest = interpolation == "nearest"
        fcs, wcs = t * h * w, (t - 1) * h * w
        kw = dict(c=c, h=h, w=w, feat_chan_stride=fcs, flow_chan_stride=wcs, nearest=nearest, coord_f16=coord_f16,
                  fuse_scale=float(fuse_scale), alpha1=float(alpha1), alpha2=float(alpha2))

        def frame(tensor, i):
            return tensor[0, :, i]                      # (C,h,w) view, channel stride T*h*w

        # backward sweep: frames T-1 .. 0, propagating with the forward flows (reference :222-228)
        outb = torch.empty_like(xin)
        outb[0, :, t - 1] = xin[0, :, t - 1]
        for idx in range(t - 2, -1, -1):
            ops.propagate_step(frame(outb, idx + 1), frame(xin, idx), frame(ff, idx), frame(fb, idx), frame(outb, idx), **kw)
        # forward sweep over the backward sweep's outputs, with the backward flows (reference :229-233)
        outf = torch.empty_like(xin)
        outf[0, :, 0] = outb[0, :, 0]
        for idx in range(1, t):
            ops.propagate_step(frame(outf, idx - 1), frame(outb, idx), frame(fb, idx - 1), frame(ff, idx - 1),
                               frame(outf, idx), **kw)
        return outf.to(x.dtype)
