"""Host-side execution helpers shared by the drop-in model classes.

Activations are channels-last fp16 matrices [B*T*H*W][C] described by a `Geom`; weights live in
ordinary nn.Parameters (so state-dict keys / `load_state_dict` / `.half()` / `.to()` behave like
the reference modules) and are re-packed lazily into the MFMA layout (`ops.pack_conv`) the first
time a module runs after its parameters changed.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops


@dataclass(frozen=True)
class Geom:
    b: int
    t: int
    h: int
    w: int

    @property
    def n_img(self):
        return self.b * self.t

    @property
    def hw(self):
        return self.h * self.w

    @property
    def rows(self):
        return self.b * self.t * self.h * self.w

    @property
    def rows_per_batch(self):
        return self.t * self.h * self.w

    def with_hw(self, h, w):
        return Geom(self.b, self.t, h, w)


class PackedCache:
    """Per-model cache of packed weights, invalidated when parameters are replaced/cast/moved."""

    def __init__(self):
        self.store = {}

    def clear(self):
        self.store.clear()

    def get(self, key, builder):
        v = self.store.get(key)
        if v is None:
            v = builder()
            self.store[key] = v
        return v


class EngineModule(nn.Module):
    """nn.Module whose packed-weight cache is dropped on _apply (half()/to()/float()) and on
    load_state_dict."""

    def _cache(self) -> PackedCache:
        c = self.__dict__.get("_uav_cache")
        if c is None:
            c = PackedCache()
            self.__dict__["_uav_cache"] = c
        return c

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        for m in self.modules():
            c = m.__dict__.get("_uav_cache")
            if c is not None:
                c.clear()
        return r

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        for m in self.modules():
            c = m.__dict__.get("_uav_cache")
            if c is not None:
                c.clear()
        return r


def _dev(p):
    if not p.is_cuda:
        raise ops._lib.UavError("model parameters must be on the GPU before forward (call .to('cuda')); "
                                "libuav_hip.so has no CPU path")
    return p.device


def packed_conv(mod: EngineModule, name, conv: nn.Module, geglu=False):
    """Packed weights of an nn.Conv2d / nn.Conv3d / nn.Linear container owned by `mod`."""
    def build():
        dev = _dev(conv.weight)
        return ops.pack_conv(conv.weight, conv.bias, geglu=geglu, device=dev)
    return mod._cache().get(("conv", name), build)


def packed_cat(mod: EngineModule, name, linears):
    """Fused projection: rows of several bias-free nn.Linear weights concatenated (q|k|v)."""
    def build():
        dev = _dev(linears[0].weight)
        w = torch.cat([l.weight.detach() for l in linears], dim=0)
        b = None
        if linears[0].bias is not None:
            b = torch.cat([l.bias.detach() for l in linears], dim=0)
        return ops.pack_conv(w, b, device=dev)
    return mod._cache().get(("cat", name), build)


def f32_param(mod: EngineModule, name, tensor):
    def build():
        _dev(tensor)
        return tensor.detach().float().contiguous()
    return mod._cache().get(("f32", name), build)


def f16_param(mod: EngineModule, name, tensor):
    def build():
        _dev(tensor)
        return tensor.detach().half().contiguous()
    return mod._cache().get(("f16", name), build)


def group_norm(mod: EngineModule, name, gn: nn.GroupNorm, x, *, n_inst, rows_per_inst, silu, x2=None, c_real=None):
    g = f32_param(mod, name + ".g", gn.weight)
    b = f32_param(mod, name + ".b", gn.bias)
    return ops.groupnorm(x, g, b, n_inst=n_inst, rows_per_inst=rows_per_inst, groups=gn.num_groups, eps=gn.eps,
                         silu=silu, x2=x2, c_real=c_real)


def layer_norm(mod: EngineModule, name, ln: nn.LayerNorm, x):
    g = f32_param(mod, name + ".g", ln.weight)
    b = f32_param(mod, name + ".b", ln.bias)
    return ops.layernorm(x, g, b, ln.eps)


def to_rows(x5, c_pad=None, x5b=None, scale=1.0):
    """(B,C,T,H,W) [+ channel-concatenated second tensor] -> channels-last rows, Geom."""
    b, c, t, h, w = x5.shape
    cc = c + (0 if x5b is None else x5b.shape[1])
    if c_pad is None:
        c_pad = (cc + 7) // 8 * 8
    if x5.dtype not in (torch.float16, torch.float32):
        x5 = x5.float()
        x5b = None if x5b is None else x5b.float()
    if x5b is not None and x5b.dtype != x5.dtype:
        x5b = x5b.to(x5.dtype)
    rows = ops.pack_nhwc(x5.contiguous(), None if x5b is None else x5b.contiguous(), c_pad=c_pad, scale=scale)
    return rows, Geom(b, t, h, w)


def from_rows(rows, g: Geom, c, out_dtype=torch.float16, clamp=None):
    return ops.unpack_ncthw(rows, c=c, n_batch=g.b, t_len=g.t, h=g.h, w=g.w, out_dtype=out_dtype, clamp=clamp)
