"""SURVEY §8 row f4, device side: the CLI's frame pre/post-processing as HIP kernels (uav/video_io.py, csrc/colorfix.hip K13)
against the CLI's own eager tensor ops (inference_upscale_a_video.py:180-187,357-359) — bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cli_pre(vframes):                      # the reference lines, verbatim semantics, on CPU
    v = (vframes / 255. - 0.5) * 2
    h, w = v.shape[-2:]
    if h >= 1280 and w >= 1280:
        v = torch.nn.functional.interpolate(v, (int(h // 4), int(w // 4)), mode='area')
    return v.unsqueeze(0).permute(0, 2, 1, 3, 4).contiguous()


def _cli_post(output):
    u = (output / 2 + 0.5).clamp(0, 1) * 255
    return u.permute(0, 2, 3, 1).contiguous().cpu().numpy().astype(np.uint8)


@pytest.mark.parametrize("shape,u8", [((3, 3, 40, 56), True), ((2, 3, 37, 51), False), ((2, 3, 1280, 1284), True)])
def test_preprocess_frames_matches_cli(dev, shape, u8):
    from uav import video_io
    g = torch.Generator().manual_seed(sum(shape))
    frames = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    ref = _cli_pre(frames.float())
    src = frames if u8 else frames.float()
    out = video_io.preprocess_frames(src.to(dev))
    assert out.shape == ref.shape and out.dtype == torch.float32
    if shape[-1] >= 1280:                   # area pooling: 16-term fp32 means, summation order may differ in the last bit
        assert (out.cpu() - ref).abs().max().item() < 2e-6
    else:
        assert torch.equal(out.cpu(), ref)


def test_postprocess_frames_matches_cli(dev):
    from uav import video_io
    g = torch.Generator().manual_seed(3)
    out = torch.randn(3, 3, 45, 70, generator=g) * 0.8
    out[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 0.999999, -0.999999, 0.5])
    ref = _cli_post(out)
    mine = video_io.postprocess_frames(out.to(dev))
    assert mine.dtype == torch.uint8 and tuple(mine.shape) == ref.shape
    assert np.array_equal(mine.cpu().numpy(), ref)
