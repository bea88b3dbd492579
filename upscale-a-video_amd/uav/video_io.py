"""Device-side halves of the reference CLI's frame I/O (`inference_upscale_a_video.py:180-187,357-359`).

Decoding / encoding video files and the LLaVA captioner stay with the CLI (host side: torchvision / imageio / transformers,
none of which the engine replaces); what runs on the GPU around `pipeline(...)` is here as two calls, each one kernel pass
instead of the CLI's chains of eager tensor ops over the whole clip (4K output: 5 passes over 3.2 GB):

    clip = preprocess_frames(frames)        # (T,C,H,W) 0..255 -> (1,C,T,h,w) fp32 in [-1,1], >= 1280-px inputs area-pooled /4
    video = postprocess_frames(output)      # (T,C,H,W) in [-1,1] -> (T,H,W,C) uint8, ready for imageio.mimwrite
"""
import torch

from . import engine as E
from . import ops


def preprocess_frames(frames: torch.Tensor) -> torch.Tensor:
    """`vframes = (vframes/255. - 0.5) * 2`; `if h >= 1280 and w >= 1280: F.interpolate(vframes, (h//4, w//4), mode='area')`;
    `unsqueeze(0)` + `rearrange('b t c h w -> b c t h w')` (inference_upscale_a_video.py:180-187)."""
    with E.device_guard(frames):
        clip = ops.frames_to_clip_f32(frames.contiguous())                      # (C,T,H,W)
        h, w = clip.shape[-2:]
        if h >= 1280 and w >= 1280:
            clip = ops.resize_area_f32(clip, int(h // 4), int(w // 4))
        return clip.unsqueeze(0)


def postprocess_frames(output: torch.Tensor) -> torch.Tensor:
    """`(output / 2 + 0.5).clamp(0, 1) * 255`, `rearrange('t c h w -> t h w c')`, `.astype(np.uint8)` (:357-359); the result
    stays on the device — `.cpu().numpy()` is the caller's (one 1-byte-per-value copy instead of an fp32 one)."""
    with E.device_guard(output):
        return ops.clip_to_frames_u8(output.float().contiguous())
