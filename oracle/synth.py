"""TEST INFRASTRUCTURE ONLY.  Deterministic synthetic weights and inputs.

No released weights exist in the reference tree (pretrained_models/.gitkeep only), so parity is
established with seeded synthetic weights.  Every parameter is drawn from a
generator seeded by crc32(parameter name) ^ base seed, so the value of a parameter depends only on
its NAME and SHAPE, not on module construction order: the reference module, the oracle
restatement and the HIP engine get bit-identical weights from their own state-dict key lists.

Zero-initialised layers of the reference (attn_temporal.to_out, TemporalModule3D.shift_conv,
ResnetBlock3D_plus.conv_3d: attention.py:490, temporal_module.py:172, resnet.py:461) are drawn
like every other layer — otherwise the temporal branches contribute exactly 0 and go untested.
All values are rounded to fp16-representable numbers (the reference runs the UNet in fp16;
the engine stores weights in fp16), so both sides compute on identical parameters.
"""
import zlib

import torch


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synth_tensor(name, shape, seed=1234):
    g = _gen(name, seed)
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    is_norm = any(s in name for s in ("norm", "group_norm")) and len(shape) == 1
    if leaf == "freqs":                      # RotaryEmbedding(dim): analytic, 1/10000^(2i/dim)
        dim = shape[0] * 2
        return (1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))).half().float()   # as in a .half() UNet
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        t = 0.5 + torch.rand(shape, generator=g)
    elif leaf == "running_mean":
        t = 0.1 * torch.randn(shape, generator=g)
    elif is_norm and leaf == "weight":
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif is_norm and leaf == "bias":
        t = 0.1 * torch.randn(shape, generator=g)
    elif "relative_attention_bias" in name:
        t = 0.5 * torch.randn(shape, generator=g)
    elif "class_embedding" in name:
        t = 0.5 * torch.randn(shape, generator=g)
    elif len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = torch.randn(shape, generator=g) * (fan_in ** -0.5)
    else:                                    # plain biases
        t = 0.1 * torch.randn(shape, generator=g)
    return t.half().float()


def synth_state_dict(shapes, seed=1234):
    """shapes: mapping name -> shape (or tensor).  Returns name -> fp32 tensor (fp16-representable)."""
    out = {}
    for k, v in shapes.items():
        shp = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        out[k] = synth_tensor(k, shp, seed)
    return out


def synth_clip(b, t, h, w, seed=0, motion=(2, 1)):
    """Low-res clip in [-1,1], (b,3,t,h,w): box-filtered noise translating by `motion` px/frame
    (trackable structure for RAFT)."""
    g = torch.Generator().manual_seed(seed)
    pad = 8 + max(abs(motion[0]), abs(motion[1])) * t
    base = torch.rand((b, 3, h + 2 * pad, w + 2 * pad), generator=g) * 2 - 1
    base = torch.nn.functional.avg_pool2d(base, 5, stride=1, padding=2)
    base = base / base.abs().max()
    frames = []
    for i in range(t):
        dx, dy = motion[0] * i, motion[1] * i
        frames.append(base[:, :, pad - dy: pad - dy + h, pad - dx: pad - dx + w])
    return torch.stack(frames, dim=2).contiguous().half().float()


def synth_prompt_embeds(prompt, dim, seq=77, seed=77):
    """Stand-in for the CLIP text encoder: deterministic (1,seq,dim)."""
    g = _gen("prompt:" + prompt, seed)
    return torch.randn((1, seq, dim), generator=g).half().float()
