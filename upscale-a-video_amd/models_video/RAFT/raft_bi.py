"""RAFT_bi — bidirectional RAFT optical flow for the flow-guided propagation (drop-in surface of the
reference's `models_video/RAFT/raft_bi.py`: RAFT_bi.forward :47-68, forward_slicing :71-104).

STATUS (round 1): the class surface and the clip-slicing schedule are in place; the RAFT network
itself (fp32 encoders, all-pairs correlation pyramid, 20 GRU iterations, convex upsampling — K11 of
SURVEY.md §7) is not built yet, so `forward` raises.  Pre-computed flows can be handed to the
pipeline through `flows_bi` (that is the only way flows enter the hot loop: pipeline :652-657), which
is how BASELINE config 3 style runs are exercised until K11 lands.  RAFT is 0.06 % of the FLOPs of a
clip and runs once per video, outside the reference's own timed region (inference :191 vs :205).
"""
import torch
import torch.nn as nn


def clip_slices(video_length, width):
    """Frame ranges of forward_slicing (reference :73-92): 12/8/4/2-frame clips by width with a
    one-frame halo on every clip but the first."""
    if width <= 640:
        n = 12
    elif width <= 720:
        n = 8
    elif width <= 1280:
        n = 4
    else:
        n = 2
    if video_length <= n:
        return [(0, video_length)]
    return [((f if f == 0 else f - 1), min(video_length, f + n)) for f in range(0, video_length, n)]


class RAFT_bi(nn.Module):
    def __init__(self, model_path="weights/raft-things.pth", device="cuda"):
        super().__init__()
        self.model_path, self._device = model_path, device
        self.eval()

    def forward(self, gt_local_frames, iters=20):
        raise NotImplementedError("RAFT (K11) is not built yet in the MI355X engine; pass precomputed `flows_bi` "
                                  "to the pipeline (see DESIGN.md, 'Not built yet')")

    def forward_slicing(self, gt_local_frames, iters=20):
        t = gt_local_frames.size(2)
        ff, fb = [], []
        for (s, e) in clip_slices(t, gt_local_frames.size(-1)):
            f, b = self.forward(gt_local_frames[:, :, s:e], iters=iters)
            ff.append(f); fb.append(b)
        return torch.cat(ff, dim=2), torch.cat(fb, dim=2)
