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


def _stamp(tensors):
    """Identity of a parameter's CURRENT value as far as autograd's bookkeeping shows it: storage pointer + the tensor's
    in-place version counter (bumped by `p.copy_`, `p.mul_`, `random_init_`, optimiser steps, load_state_dict on a plain
    nn.Linear child ...); replaced / cast / moved parameters get a new pointer.  NOT covered: writes through `p.data`
    (`p.data.copy_(w)` — `.data` is a detached alias with its OWN version counter, `p._version` stays put) and raw-pointer
    writes; after those call `invalidate_packed(model)`."""
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


def invalidate_packed(model: nn.Module):
    """Drop every packed-weight / table cache under `model` (all EngineModule children).  Needed after weight edits the
    stamp cannot see: `p.data.copy_(...)`, `p.data.normal_()`, writes from another framework through the data pointer."""
    for m in model.modules():
        c = m.__dict__.get("_uav_cache")
        if c is not None:
            c.clear()
        m.__dict__.pop("_ehs_src", None)
    return model


class PackedCache:
    """Per-model cache of packed weights.  Dropped wholesale on _apply / load_state_dict / `invalidate_packed`, and every
    entry carries the stamp of the parameters it was built from, so in-place edits of the PARAMETER after the first
    forward (`p.copy_`, `init_weights.random_init_`, load_state_dict on a plain nn.Linear child) rebuild it instead of
    serving stale packed fp16 weights.  Edits through `p.data` are invisible to the stamp (see `_stamp`)."""

    def __init__(self):
        self.store = {}

    def clear(self):
        self.store.clear()

    def get(self, key, builder, src=()):
        stamp = _stamp(src)
        hit = self.store.get(key)
        if hit is None or hit[0] != stamp:
            hit = (stamp, publish(builder()))
            self.store[key] = hit
        return acquire(hit[1])


# Shared, lazily built device objects (packed weights, tables, text K/V, prompt rows) are produced on whatever stream first needs
# them and then used from any stream (the two clips of the serving mode, the overlapped windows / decode chunks of `uav.streams`).
# Ordering without a host rendezvous: the builder records an EVENT behind the kernels that fill the object
# (`publish(obj)`); until that event has completed, every user makes its own current stream wait for it (`acquire(obj)`) — a
# device-side dependency, the host never blocks.  `_PENDING` is empty in the steady state, so the check is one dict truth test.
_PENDING = {}          # id(obj) -> (obj, event): objects whose build may still be in flight


def publish(obj=None):
    """Call right after the kernels that fill the shared object `obj` were issued on the current stream."""
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        return obj
    if obj is None:                                  # no handle to hang the event on: host-wait (legacy form)
        torch.cuda.current_stream().synchronize()
        return obj
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    # retire the builds whose event has completed on EVERY publish (the dict holds a strong reference to the object — an evicted
    # text-K/V tuple or class-label index would otherwise stay pinned in device memory until it happened to be acquired again; the
    # dict is a handful of entries in the steady state).  Host threads of the two-clip mode share the dict: every
    # operation here is a single dict call under the GIL, and a doubly-popped key is harmless (pop(k, None)).
    for k in [k for k, (_, e) in list(_PENDING.items()) if e.query()]:
        _PENDING.pop(k, None)
    _PENDING[id(obj)] = (obj, ev)
    return obj


def acquire(obj):
    """Call before using a shared object that some stream may still be building; returns `obj`."""
    if _PENDING:
        hit = _PENDING.get(id(obj))
        if hit is not None and hit[0] is obj:
            if hit[1].query():
                _PENDING.pop(id(obj), None)
            else:
                torch.cuda.current_stream().wait_event(hit[1])
    return obj


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


def device_guard(t):
    """Context manager making `t`'s GPU the current device for the duration of a model call: launches go to torch's
    current stream of the current device (ops._stream), so a model sitting on a non-current GPU (`pipe.to('cuda:1')`
    without torch.cuda.set_device) would otherwise launch on device 0's stream against device-1 pointers."""
    if isinstance(t, torch.Tensor) and t.is_cuda:
        return torch.cuda.device(t.device)
    import contextlib
    return contextlib.nullcontext()


def guarded(fn):
    """Method decorator: run `fn` with the GPU of its first CUDA tensor argument as the current device."""
    import functools

    @functools.wraps(fn)
    def wrap(self, *a, **kw):
        t = next((x for x in list(a) + list(kw.values()) if isinstance(x, torch.Tensor) and x.is_cuda), None)
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(self, *a, **kw)
        with torch.cuda.device(t.device):
            return fn(self, *a, **kw)
    return wrap


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
    return mod._cache().get(("conv", name), build, (conv.weight, conv.bias))


def packed_conv_hilo(mod: EngineModule, name, conv: nn.Module):
    """1x1 conv whose operand arrives as [hi | lo] fp16 halves of an fp32 row (ops.groupnorm_apply want_raw="hilo"): the weights
    repeated along K, W.[hi | lo] = W.hi + W.lo.  `cin` keeps the logical channel count (FLOP accounting: the second half is
    implementation overhead, not algorithmic work)."""
    def build():
        dev = _dev(conv.weight)
        w = conv.weight.detach()
        if any(k != 1 for k in w.shape[2:]):
            raise ops._lib.UavError("hi/lo operands are for 1x1 convs")
        cw = ops.pack_conv(torch.cat([w, w], dim=1), conv.bias, device=dev)
        cw.cin = w.shape[1]
        return cw
    return mod._cache().get(("conv_hilo", name), build, (conv.weight, conv.bias))


def packed_conv_dup(mod: EngineModule, name, conv: nn.Module):
    """Any conv whose operand arrives as [hi | lo] halves per pixel (2*C_in channels): the weights repeated along C_in."""
    def build():
        dev = _dev(conv.weight)
        w = conv.weight.detach()
        cw = ops.pack_conv(torch.cat([w, w], dim=1), conv.bias, device=dev)
        cw.cin = w.shape[1]
        return cw
    return mod._cache().get(("conv_dup", name), build, (conv.weight, conv.bias))


def hilo_rows(x):
    """fp32 rows [M][C] -> fp16 rows [M][2C] = [fp16(x) | fp16(x - fp16(x))]."""
    return ops.cast_hilo(x)


def packed_conv_with_shortcut(mod: EngineModule, name, conv: nn.Module, shortcut: nn.Module, repeat_short):
    """conv2 of a ResNet block with its 1x1 shortcut conv folded in as extra K (ops.pack_conv_with_shortcut)."""
    def build():
        dev = _dev(conv.weight)
        return ops.pack_conv_with_shortcut(conv.weight, conv.bias, shortcut.weight, shortcut.bias, repeat_short, device=dev)
    return mod._cache().get(("conv+shortcut", name, repeat_short), build, (conv.weight, conv.bias, shortcut.weight, shortcut.bias))


def packed_upsample_phases(mod: EngineModule, name, conv: nn.Module, dup=False):
    """The four 2x2 sub-pixel phase convs of an upsampler's 3x3 conv, packed ([py][px], see ops.upsample_phase_weights)."""
    def build():
        dev = _dev(conv.weight)
        ph = ops.upsample_phase_weights(conv.weight)
        cin = ph[0][0].shape[1]
        if dup:
            ph = [[torch.cat([ph[py][px]] * 2, dim=1) for px in range(2)] for py in range(2)]
        cws = [[ops.pack_conv(ph[py][px], conv.bias, device=dev) for px in range(2)] for py in range(2)]
        for row in cws:
            for cw in row:
                cw.cin = cin          # logical channels (FLOP accounting: the lo half is implementation overhead, not algorithmic work)
        return cws
    return mod._cache().get(("up_phases", name, dup), build, (conv.weight, conv.bias))


def packed_cat(mod: EngineModule, name, linears):
    """Fused projection: rows of several bias-free nn.Linear weights concatenated (q|k|v)."""
    def build():
        dev = _dev(linears[0].weight)
        w = torch.cat([l.weight.detach() for l in linears], dim=0)
        b = None
        if linears[0].bias is not None:
            b = torch.cat([l.bias.detach() for l in linears], dim=0)
        return ops.pack_conv(w, b, device=dev)
    return mod._cache().get(("cat", name), build, [l.weight for l in linears] + [l.bias for l in linears])


def f32_param(mod: EngineModule, name, tensor):
    def build():
        _dev(tensor)
        return tensor.detach().float().contiguous()
    return mod._cache().get(("f32", name), build, (tensor,))


def f16_param(mod: EngineModule, name, tensor):
    def build():
        _dev(tensor)
        return tensor.detach().half().contiguous()
    return mod._cache().get(("f16", name), build, (tensor,))


# fp32-stream blocks (VAE decoder always; UNet with stream_dtype = float32): is the ResNet BRANCH tensor between conv1 and
# norm2 kept in fp32 as well (1), or rounded to fp16 like an MFMA operand (0)?  UAV_BRANCH_F32.
import os as _os
# Round 3 measurement (GPU, full width): fp16 costs 8.07e-4 -> 8.55e-4 per forward, 7.7e-4 -> 8.1e-4 at the headline shape, 8.6e-4 ->
# 8.8e-4 after 30 steps, 5.8e-4 -> 6.1e-4 for a decode chunk, and saves 2.2 % of the clip time (one fp32 write + one fp32
# read of every ResNet's branch tensor): default 0.
BRANCH_F32 = _os.environ.get("UAV_BRANCH_F32", "0") != "0"
# ... and is the TOKEN stream inside a Transformer3DModel (proj_in output, the four residual adds of the block) fp32 (1) or
# fp16 (0: only the block stream around the transformer is fp32)?  UAV_TOKEN_F32, default 1.
TOKEN_F32 = _os.environ.get("UAV_TOKEN_F32", "1") != "0"
# ... only in the transformers with at most this many tokens per frame (0 = everywhere): the token stream of the
# highest-resolution transformers carries 3/4 of the token traffic.  UAV_TOKEN_F32_MAX_HW.
TOKEN_F32_MAX_HW = int(_os.environ.get("UAV_TOKEN_F32_MAX_HW", "0"))
# ... and does the 1x1 shortcut conv of a block with C_in != C_out read the fp32 stream as TWO fp16 operands (hi + lo, K
# doubled: the stream is not rounded to fp16 on its way through the block, 1.01e-3 -> 0.81e-3 per forward for ~2 % of the
# clip time) or as one (0)?  UAV_SHORTCUT_HILO, default 1.
SHORTCUT_HILO = _os.environ.get("UAV_SHORTCUT_HILO", "1") != "0"
# ... and is that shortcut conv folded into the block's conv2 (one implicit GEMM over K = 9*C + C_in: no separate launch, no fp32
# shortcut tensor written and read back as the residual) (1) or a launch of its own (0)?  UAV_FUSE_SHORTCUT, default 1.
FUSE_SHORTCUT = _os.environ.get("UAV_FUSE_SHORTCUT", "1") != "0"
# ... and do the down / up SAMPLER convs (3x3 stride 2; nearest-2x + 3x3 as four 2x2 phase convs), the other place where the
# stream itself is an MFMA operand, read it as the same hi + lo pair (K doubled on 6 convs per forward)?  UAV_SAMPLER_HILO,
# default 1 since round 4: measured at the headline shape (8 x 320x320, 30 steps, vs the GPU oracle,
# profiles/r04_parity_precision_knobs_at_headline_shape_run1.jsonl) latents 9.0e-4 -> 8.2e-4 and `.images` (all pixels)
# 1.075e-3 -> 9.5e-4, i.e. what puts the OUTPUT of the pipeline call inside the stated 1e-3.
# "down" / "up" restrict it to one kind (numerics experiments).
# ... and the block TAILS — the last feed-forward output of a Transformer3DModel (operand of proj_out) and the tail ResNet of a
# TemporalModule3D (operand of shift_conv), fp16 tensors that carry a whole residual sum — as fp32 rows read through the same
# hi | lo pair by their 1x1 consumer?  UAV_TAIL_HILO (CPU emulation: 7.5e-4 -> 6.5e-4 per forward).
# Default 1 since round 6: the producer's epilogue writes the pair itself (UAV_CONV_OUT_HILO: no cast pass), which leaves the doubled K of
# the 1x1 consumers as the only cost — measured at the headline shape (8 x 320x320, 30 steps, vs the GPU oracle, run 4 of round 6): latents
# 8.2e-4 -> 7.3e-4, `.images` 9.5e-4 -> 8.7e-4 over all pixels / 1.23e-3 -> 1.12e-3 unclamped, for -1.2 % frames/s (1.1197 -> 1.1066 same box).
TAIL_HILO = _os.environ.get("UAV_TAIL_HILO", "1") != "0"
# samplers left on a single fp16 operand although SAMPLER_HILO is on: ("up" | "down", input height) pairs (numerics experiments)
import threading as _threading

_PRECISION = _threading.local()        # .high: depth of `precision_high()` scopes open on THIS host thread


class precision_high:
    """Scope in which THIS host thread's launches lift the two cheap roundings (`UNetVideoModel.precision = "high"`): block
    tails as hi | lo pairs and the fp32 ResNet branch tensor.  Thread-local by design: one UNet is shared by the
    host threads of the two-clip mode and of `uav.streams`, so the switch must not be a process global that one thread's
    `finally` clears under another thread's forward."""

    def __enter__(self):
        _PRECISION.high = getattr(_PRECISION, "high", 0) + 1
        return self

    def __exit__(self, *exc):
        _PRECISION.high -= 1
        return False


def tail_hilo():
    """Block tails read as an fp16 hi | lo pair by their 1x1 consumer?  (UAV_TAIL_HILO, or inside `precision_high()`.)"""
    return TAIL_HILO or getattr(_PRECISION, "high", 0) > 0


def branch_f32():
    """ResNet branch tensor conv1 -> norm2 kept in fp32?  (UAV_BRANCH_F32, or inside `precision_high()`.)"""
    return BRANCH_F32 or getattr(_PRECISION, "high", 0) > 0


SAMPLER_HILO_SKIP = set()
SAMPLER_HILO = {"0": False, "1": True}.get(_os.environ.get("UAV_SAMPLER_HILO", "1"), _os.environ.get("UAV_SAMPLER_HILO", "1"))


# Group count a conv assumes for the GroupNorm that (probably) consumes its output when the caller cannot name that
# consumer (up / down samplers, TemporalModule3D.shift_conv, the decoder's conv_in): every GroupNorm of the released
# configs has 32 groups (norm_num_groups / resnet_groups).  A miss only leaves the partials unused (ops.conv_gemm).
GN_GROUPS_HINT = 32


def group_norm(mod: EngineModule, name, gn: nn.GroupNorm, x, *, n_inst, rows_per_inst, silu, x2=None, c_real=None,
               want_raw=False):
    g = f32_param(mod, name + ".g", gn.weight)
    b = f32_param(mod, name + ".b", gn.bias)
    return ops.groupnorm(x, g, b, n_inst=n_inst, rows_per_inst=rows_per_inst, groups=gn.num_groups, eps=gn.eps,
                         silu=silu, x2=x2, c_real=c_real, want_raw=want_raw)


# LayerNorm folded into the projection that consumes it (fp32 token stream only): LN(x).W^T + b = rstd*(x16.(W o gamma)^T -
# mu*colsum) + (W.beta + b); the linear that produced x writes x16 and the row statistics (ops.LnOperand), the LayerNorm pass
# (4 B/elem read + 2 written) disappears.  Built, tested and MEASURED in round 3 (profiles/r03_ab_layernorm_folded_*,
# r03_parity_layernorm_fold_*): +1.6 % frames/s (LayerNorm 209 ms gone, conv +42 ms), but the un-normalised fp16 operand costs
# accuracy where the row mean is not small against its spread: 8.5e-4 -> 8.8e-4 per forward and 8.8e-4 -> 9.5e-4 after the 30-step
# schedule — too close to the stated 1e-3.  OFF by default; UAV_LN_FOLD=1 turns it on.
LN_FOLD = _os.environ.get("UAV_LN_FOLD", "0") != "0"


# Text cross-attention sub-layers of the 512-channel levels (LayerNorm -> to_q -> 77-key softmax -> to_out -> + residual) as ONE
# launch (csrc/xattn_fused.hip, round 6): the fp32 token stream is read once and written once instead of 24 B per element over four
# launches.  UAV_XATTN_FUSED=0 keeps the four-launch chain (A/B and the equivalence test).
XATTN_FUSED = _os.environ.get("UAV_XATTN_FUSED", "1") != "0"
# ... and attn1 + attn2 of a block with only_cross_attention (both text cross-attention) as ONE launch of that kernel: the second LayerNorm
# runs on the accumulators.  UAV_XATTN_PAIR=0 keeps one launch per sub-layer.
XATTN_PAIR = _os.environ.get("UAV_XATTN_PAIR", "1") != "0"
# ... and the TEMPORAL attention sub-layer of the same blocks (LayerNorm -> q | k | v -> RoPE + relative-position bias + softmax over the 8
# frames of a pixel -> to_out -> + residual) as one launch (tattn_sublayer_kernel).  UAV_TATTN_FUSED=0 keeps the four-launch chain.
TATTN_FUSED = _os.environ.get("UAV_TATTN_FUSED", "1") != "0"
# ... and all three (attn1 -> attn2 -> attn_temporal) as ONE launch where both of the above apply (tattn_sublayer_kernel<2>).
# UAV_BLOCK_ATTN_FUSED=0 keeps the pair launch + the temporal launch.
BLOCK_ATTN_FUSED = _os.environ.get("UAV_BLOCK_ATTN_FUSED", "1") != "0"


# ... and the feed-forward sub-layer behind them (LayerNorm -> GEGLU 512 -> 2 x 2048 -> Linear 2048 -> 512 -> + residual) as one launch
# (ff_sublayer_kernel: the hidden activations never leave the registers).  UAV_FF_FUSED=0 keeps LayerNorm + the two conv-GEMM launches.
FF_FUSED = _os.environ.get("UAV_FF_FUSED", "1") != "0"
# ... and the four sub-layers of such a block in ONE launch (tattn_sublayer_kernel<2, 1>).  UAV_BLOCK_FF_FUSED=0 keeps the attention launch +
# the feed-forward launch.
BLOCK_FF_FUSED = _os.environ.get("UAV_BLOCK_FF_FUSED", "1") != "0"
# ... and GroupNorm apply -> proj_in of the Transformer3DModel in front of it in the same launch (tattn_sublayer_kernel<2, 1, 1>: from the
# GroupNorm's input to the block's output).  UAV_PROJ_IN_FUSED=0 keeps the GroupNorm-apply pass and the proj_in launch.
PROJ_IN_FUSED = _os.environ.get("UAV_PROJ_IN_FUSED", "1") != "0"
# ... and (where the feed-forward is NOT fused) the LayerNorm in front of the feed-forward (norm3) written by that launch's epilogue (fp16 rows beside the fp32 ones): the
# LayerNorm pass of the block's last sub-layer disappears.  UAV_NEXT_LN=0 keeps the pass.
NEXT_LN = _os.environ.get("UAV_NEXT_LN", "1") != "0"


def packed_ln_linear(mod: EngineModule, name, ln: nn.LayerNorm, linears, geglu=False):
    """(ConvW of fp16(W o gamma) with bias W.beta + b, colsum[n_pad] fp32) for the row-concatenated `linears` behind `ln`."""
    def build():
        dev = _dev(linears[0].weight)
        w = torch.cat([l.weight.detach().float() for l in linears], dim=0)                 # (N, K)
        g, be = ln.weight.detach().float().to(w.device), ln.bias.detach().float().to(w.device)
        b = torch.zeros(w.shape[0], device=w.device)
        if linears[0].bias is not None:
            b = torch.cat([l.bias.detach().float() for l in linears], dim=0)
        cw = ops.pack_conv(w * g[None, :], b + w @ be, geglu=geglu, device=dev)
        colsum = cw.w.float().sum(dim=1).contiguous()                                       # of the PACKED (fp16-rounded, row-permuted) weights
        return cw, colsum
    src = [l.weight for l in linears] + [l.bias for l in linears] + [ln.weight, ln.bias]
    return mod._cache().get(("ln_linear", name, geglu), build, src)


def ln_linear(mod: EngineModule, name, ln: nn.LayerNorm, x, linears, geglu=False):
    """linear(LayerNorm(x)) for the (row-concatenated) `linears`: folded when x carries an ops.LnOperand and the launch
    qualifies, else the LayerNorm pass followed by the plain (cached) projection."""
    op = ops.ln_operand_of(x) if (LN_FOLD and x.dtype == torch.float32) else None
    if op is not None:
        cw, colsum = packed_ln_linear(mod, name, ln, linears, geglu)
        if ops.ln_fold_ok(x.shape[0], x.shape[-1], cw):
            return ops.linear(op.raw, cw, ln_consume=(op, colsum, ln.eps))
    n = None
    if NEXT_LN and x.dtype == torch.float32:       # the kernel that wrote x also wrote LayerNorm(x) for THIS LayerNorm (ops.NextLn)?
        n = ops.next_ln_of(x, f32_param(mod, name + ".ln.g", ln.weight), f32_param(mod, name + ".ln.b", ln.bias), ln.eps)
    if n is None:
        n = layer_norm(mod, name + ".ln", ln, x)
    if len(linears) == 1:
        return ops.linear(n, packed_conv(mod, name, linears[0], geglu=geglu))
    return ops.linear(n, packed_cat(mod, name, linears))


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
