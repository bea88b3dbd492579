"""MI355X-native colour transfer post-processes of the CLI (`--color_fix AdaIn | Wavelet`; drop-in for the reference's
`models_video/color_correction.py`: calc_mean_std :43-57, adaptive_instance_normalization :59-71, wavelet_blur :73-91,
wavelet_decomposition :93-106, wavelet_reconstruction :108-118).

They run once per video on the decoded (T,3,4H,4W) fp32 frames — HBM-bound elementwise / 3x3-stencil work on 1280x1280
images — as HIP kernels of libuav_hip.so (csrc/colorfix.hip, K12): a deterministic per-plane statistics pass
(mean, unbiased variance) + one apply pass for AdaIN; one fused launch per a-trous level (blur + running
high-frequency sum) for the 5-level wavelet split.  `upsample_bicubic4` is the CLI's
`F.interpolate(vframes, scale_factor=4, mode='bicubic')` (inference_upscale_a_video.py:327) on the same library.
There is no CPU path: CPU tensors raise.
"""
import torch

from uav import engine as E
from uav import ops


def _f32(x):
    return x.float().contiguous()


def calc_mean_std(feat, eps=1e-5):
    """(B,C,H,W) -> per-(b,c) mean and std = sqrt(unbiased var + eps), shaped (B,C,1,1) like the reference."""
    size = feat.size()
    assert len(size) == 4, "The input feature should be 4D tensor."
    b, c = size[:2]
    with E.device_guard(feat):
        mean, var = ops.plane_stats_f32(_f32(feat))
    return mean.reshape(b, c, 1, 1), (var + eps).sqrt().reshape(b, c, 1, 1)


def adaptive_instance_normalization(content_feat, style_feat):
    """Give `content_feat` (B,C,H,W) the per-channel mean / std of `style_feat`."""
    with E.device_guard(content_feat):
        c, s = _f32(content_feat), _f32(style_feat)
        c_mean, c_var = ops.plane_stats_f32(c)
        s_mean, s_var = ops.plane_stats_f32(s)
        return ops.adain_apply_f32(c, c_mean, c_var, s_mean, s_var, eps=1e-5).to(content_feat.dtype)


def adain_color_fix(target_tensor, source_tensor):
    return adaptive_instance_normalization(target_tensor, source_tensor)


def wavelet_blur(image, radius):
    with E.device_guard(image):
        return ops.atrous_blur_f32(_f32(image), radius).to(image.dtype)


def wavelet_decomposition(image, levels=5):
    """(high, low) of the 5-level a-trous split: low = blur_16(...blur_1(image)), high = sum_i (image_i - low_i)."""
    with E.device_guard(image):
        cur = _f32(image)
        high = torch.zeros_like(cur)
        for i in range(levels):
            cur = ops.atrous_blur_f32(cur, 2 ** i, high=high)
        return high.to(image.dtype), cur.to(image.dtype)


def wavelet_reconstruction(content_feat, style_feat):
    """High frequencies of `content_feat` + low frequencies (colour) of `style_feat`."""
    with E.device_guard(content_feat):
        high, _ = wavelet_decomposition(_f32(content_feat))
        low = _f32(style_feat)
        for i in range(5):                       # the style's high frequencies are never needed
            low = ops.atrous_blur_f32(low, 2 ** i)
        return ops.axpby_f32(high, low, 1.0, 1.0).to(content_feat.dtype)


def wavelet_color_fix(target_tensor, source_tensor):
    return wavelet_reconstruction(target_tensor, source_tensor)


def upsample_bicubic4(frames):
    """F.interpolate(frames, scale_factor=4, mode='bicubic') on (T,C,H,W) frames (inference_upscale_a_video.py:327)."""
    with E.device_guard(frames):
        h, w = frames.shape[-2:]
        return ops.resize_bicubic_f32(_f32(frames), 4 * h, 4 * w, scale_h=0.25, scale_w=0.25).to(frames.dtype)
