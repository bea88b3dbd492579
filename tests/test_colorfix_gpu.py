"""Colour-correction kernels (K12, csrc/colorfix.hip) through the drop-in `models_video.color_correction` functions,
against the outputs of the REFERENCE's own module (tests/golden/colorfix.pt, oracle/make_golden.py --colorfix) and the
oracle restatement.  fp32 in / fp32 out: tolerance 2e-5 absolute on values in [-1.6, 1.6] (summation order of the
stencil / statistics differs from ATen's)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


def maxabs(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def test_colorfix_vs_reference(dev):
    import golden_cases as GC
    import uav_oracle as O
    from models_video import color_correction as CC
    lr, content = GC.colorfix_inputs()
    gold = torch.load(os.path.join(GOLD, "colorfix.pt"))
    style = CC.upsample_bicubic4(lr.to(dev))
    assert style.shape == gold["style_bicubic4"].shape and maxabs(style, gold["style_bicubic4"]) < 2e-5
    assert maxabs(style, O.bicubic4(lr)) < 2e-5
    ad = CC.adaptive_instance_normalization(content.to(dev), gold["style_bicubic4"].to(dev))
    assert maxabs(ad, gold["adain"]) < 2e-5 and maxabs(ad, O.adain(content, gold["style_bicubic4"])) < 2e-5
    wv = CC.wavelet_reconstruction(content.to(dev), gold["style_bicubic4"].to(dev))
    assert maxabs(wv, gold["wavelet"]) < 2e-5
    high, low = CC.wavelet_decomposition(content.to(dev))
    assert maxabs(high, gold["high"]) < 2e-5 and maxabs(low, gold["low"]) < 2e-5
    mean, std = CC.calc_mean_std(content.to(dev))
    assert mean.shape == (2, 3, 1, 1)
    assert maxabs(mean, content.mean(dim=(2, 3), keepdim=True)) < 1e-6
    assert maxabs(std, (content.var(dim=(2, 3), keepdim=True) + 1e-5).sqrt()) < 1e-6


def test_colorfix_full_frame_properties(dev):
    """1280x1280 frames (config 2's output size): AdaIN output carries exactly the style's per-plane mean / std; the
    wavelet split is a partition (high + low = image); a constant image is a fixed point of the blur at every level."""
    from models_video import color_correction as CC
    gd = torch.Generator(device=dev).manual_seed(3)
    content = torch.rand(2, 3, 1280, 1280, generator=gd, device=dev) * 1.6 - 0.8
    style = torch.rand(2, 3, 1280, 1280, generator=gd, device=dev) * 0.5 + 0.1
    out = CC.adaptive_instance_normalization(content, style)
    om, osd = CC.calc_mean_std(out, eps=0.0)
    sm, ssd = CC.calc_mean_std(style, eps=1e-5)          # output std = sqrt(style var + eps) * content std / sqrt(content var + eps)
    assert maxabs(om, sm) < 1e-5 and maxabs(osd, ssd) < 1e-4
    high, low = CC.wavelet_decomposition(content)
    assert maxabs(high + low, content) < 1e-5
    const = torch.full((1, 3, 300, 200), 0.37, device=dev)
    for r in (1, 2, 4, 8, 16):
        assert maxabs(CC.wavelet_blur(const, r), const) < 1e-6
    assert torch.equal(CC.wavelet_reconstruction(content, style), CC.wavelet_reconstruction(content, style))   # deterministic


def test_colorfix_has_no_cpu_path():
    from models_video import color_correction as CC
    from uav import _lib
    with pytest.raises(_lib.UavError):
        CC.adaptive_instance_normalization(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))
