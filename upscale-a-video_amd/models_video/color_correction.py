"""Colour transfer post-processes of the CLI (`--color_fix adain|wavelet`, default None).

OUT OF THE HOT-PATH SCOPE (SURVEY.md §2 row 8, §8f "next" item 3): these run once per video on the
decoded frames and are NOT MI355X-native yet — plain tensor math kept only so that the reference
CLI's import line (inference_upscale_a_video.py:45) resolves against this package.  Semantics follow
the reference's `models_video/color_correction.py` (:59-71 AdaIN statistics, :73-118 five-level
a-trous wavelet low/high split).
"""
import torch
import torch.nn.functional as F


def _stats(feat, eps=1e-5):
    b, c = feat.shape[:2]
    flat = feat.reshape(b, c, -1)
    return flat.mean(dim=2).reshape(b, c, 1, 1), (flat.var(dim=2) + eps).sqrt().reshape(b, c, 1, 1)


def adaptive_instance_normalization(content_feat, style_feat):
    """Give `content_feat` (B,C,H,W) the per-channel mean / std of `style_feat`."""
    s_mean, s_std = _stats(style_feat)
    c_mean, c_std = _stats(content_feat)
    return (content_feat - c_mean) / c_std * s_std + s_mean


_BLUR = ((0.0625, 0.125, 0.0625), (0.125, 0.25, 0.125), (0.0625, 0.125, 0.0625))


def _atrous_blur(image, radius):
    k = torch.tensor(_BLUR, dtype=image.dtype, device=image.device)[None, None].repeat(image.shape[1], 1, 1, 1)
    padded = F.pad(image, (radius,) * 4, mode="replicate")
    return F.conv2d(padded, k, groups=image.shape[1], dilation=radius)


def _split(image, levels=5):
    high = torch.zeros_like(image)
    for i in range(levels):
        low = _atrous_blur(image, 2 ** i)
        high = high + (image - low)
        image = low
    return high, image


def wavelet_reconstruction(content_feat, style_feat):
    """High frequencies of `content_feat` + low frequencies (colour) of `style_feat`."""
    high, _ = _split(content_feat)
    _, low = _split(style_feat)
    return high + low
