"""RAFT_bi — bidirectional RAFT optical flow for the flow-guided propagation (drop-in for the
reference's `models_video/RAFT/raft_bi.py`: initialize_RAFT :19-33, RAFT_bi.forward :47-68,
forward_slicing :71-104).  fp32, like the reference.  The trilinear pre-resize / bilinear flow
resize of the reference (:53,:62-63) run on `uav_resize_bilinear_f32` when H or W is not a multiple
of 8 (they are identities otherwise and skipped).
"""
import torch
import torch.nn as nn

from uav import engine as E
from uav import ops

from .raft import RAFT


def clip_slices(video_length, width):
    """Frame ranges of forward_slicing (reference :73-92): 12/8/4/2-frame clips by width with a
    one-frame halo on every clip but the first."""
    n = 12 if width <= 640 else 8 if width <= 720 else 4 if width <= 1280 else 2
    if video_length <= n:
        return [(0, video_length)]
    return [((f if f == 0 else f - 1), min(video_length, f + n)) for f in range(0, video_length, n)]


def initialize_RAFT(model_path="pretrained_models/raft-things.pth", device="cuda"):
    model = RAFT()
    if model_path is not None:
        sd = torch.load(model_path, map_location="cpu")
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}     # saved from nn.DataParallel
        model.load_state_dict(sd)
    return model.to(device)


class RAFT_bi(nn.Module):
    def __init__(self, model_path="weights/raft-things.pth", device="cuda"):
        super().__init__()
        self.fix_raft = initialize_RAFT(model_path, device=device)
        for p in self.fix_raft.parameters():
            p.requires_grad = False
        self.eval()

    @E.guarded
    def forward(self, gt_local_frames, iters=20):
        b, c, t, h, w = gt_local_frames.size()
        h8, w8 = -(-h // 8) * 8, -(-w // 8) * 8
        frames = gt_local_frames
        if (h8, w8) != (h, w):
            # reference :53 — trilinear to (T, H_, W_); T is unchanged, so it is a per-frame bilinear resize
            frames = ops.resize_bilinear_f32(frames.float(), h8, w8)
        ff, fb = [], []
        for i in range(b):
            f, bwd = self.fix_raft.flows_bidirectional(frames[i].permute(1, 0, 2, 3).contiguous(), iters=iters)
            if (h8, w8) != (h, w):
                # resize_flow_pytorch (:11-16), including its row-indexed rescale (`flow[:, :, 0] *= newh/oldh`)
                f = ops.resize_bilinear_f32(f, h, w, row0_scale=h / h8, row1_scale=w / w8)
                bwd = ops.resize_bilinear_f32(bwd, h, w, row0_scale=h / h8, row1_scale=w / w8)
            ff.append(f.permute(1, 0, 2, 3)); fb.append(bwd.permute(1, 0, 2, 3))
        return torch.stack(ff).contiguous(), torch.stack(fb).contiguous()

    def forward_slicing(self, gt_local_frames, iters=20):
        t = gt_local_frames.size(2)
        ff, fb = [], []
        for (s, e) in clip_slices(t, gt_local_frames.size(-1)):
            f, b = self.forward(gt_local_frames[:, :, s:e], iters=iters)
            ff.append(f); fb.append(b)
        return torch.cat(ff, dim=2), torch.cat(fb, dim=2)
