"""Random-init helper for benchmarks / smoke runs (no released weights exist offline).

Same rules as oracle/synth.py (fan-in scaled normal for conv/linear, norm gains ~ N(1,0.1),
zero-initialised temporal layers re-drawn so they carry signal) but drawn directly on the device
with one generator — values need to be sane, not reproducible across hosts.
"""
import torch


@torch.no_grad()
def random_init_(model: torch.nn.Module, seed: int = 1234, device=None):
    dev = device or next(model.parameters()).device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    for name, p in list(model.named_parameters()) + list(model.named_buffers()):
        leaf = name.split(".")[-1]
        if leaf == "freqs" or not p.is_floating_point():
            continue
        shape = tuple(p.shape)
        is_norm = ("norm" in name) and len(shape) == 1
        if is_norm and leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=dev)
        elif is_norm or len(shape) < 2:
            t = 0.1 * torch.randn(shape, generator=g, device=dev)
        elif "relative_attention_bias" in name or "class_embedding" in name:
            t = 0.5 * torch.randn(shape, generator=g, device=dev)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g, device=dev) * (fan_in ** -0.5)
        p.copy_(t.to(p.dtype))
    from . import engine
    engine.invalidate_packed(model)          # explicit: packed fp16 copies of the old values must not survive
    return model
