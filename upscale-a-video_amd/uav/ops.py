"""Tensor-level wrappers over the C ABI (include/uav_hip.h).

PyTorch is used for device memory and streams only: every function takes CUDA(=HIP) tensors,
passes `data_ptr()`s and the current stream to libuav_hip.so and returns torch tensors that own
the output buffers.  Activations are channels-last fp16 matrices [rows][C], rows ordered
(batch, frame, y, x).  There is no eager/CPU fallback.
"""
import ctypes as C
import functools
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib

HALF = torch.float16


import threading

_TLS = threading.local()     # .dev: device index of the tensor this THREAD most recently handed to _p()


def _stream():
    """Stream every launch of this call is ordered on: torch's current stream of the CURRENT device.  The tensors whose
    pointers were just taken (_p) must live on that device — kernels launched on device A's stream against device-B
    pointers fault or run on the wrong GPU — so a mismatch raises instead (the model entry points switch the current
    device to their tensors' device, engine.device_guard; library state such as the dynamic-LDS attribute is per device)."""
    cur = torch.cuda.current_device()
    last = getattr(_TLS, "dev", -1)
    if last != cur:
        raise _lib.UavError(f"tensors live on cuda:{last} but the current device is cuda:{cur}: wrap the call in "
                            f"`with torch.cuda.device({last})` (the model / pipeline entry points do)")
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    _TLS.dev = t.device.index
    return t.data_ptr()


def _req(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise _lib.UavError(f"{name} must live on the GPU (libuav_hip.so has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.UavError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.UavError(f"{name} must be contiguous")
    return t


class KernelProfiler:
    """Optional per-launch HIP-event timing (bench.py roofline leg).  Events are recorded on torch's
    current stream, which is the stream every libuav_hip.so launch is issued on."""

    def __init__(self):
        self.enabled = False
        self.detail = False        # key conv launches by shape (tools / UAV_BENCH_DETAIL)
        self.only = None           # set of kernel families to time (None = all); each event pair costs ~3 us of GPU time
        self.records = []          # (kernel, flops, bytes, start_event, end_event)

    def start(self, only=None):
        self.enabled, self.records, self.only = True, [], (None if only is None else set(only))

    def stop(self):
        self.enabled = False

    def begin(self, family=None):
        if not self.enabled or (self.only is not None and family not in self.only):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, e0, kernel, flops=0.0, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((kernel, flops, nbytes, e0, e1))

    def summary(self):
        """kernel -> dict(launches, seconds, flops, bytes); call after a device synchronize."""
        out = {}
        for k, fl, nb, e0, e1 in self.records:
            d = out.setdefault(k, dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1; d["seconds"] += e0.elapsed_time(e1) * 1e-3; d["flops"] += fl; d["bytes"] += nb
        return out


PROFILER = KernelProfiler()

_ZERO = {}


def zero_page(device):
    key = str(device)
    if key not in _ZERO:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        if z.is_cuda:
            torch.cuda.current_stream(z.device).synchronize()      # shared by every stream from here on
        _ZERO[key] = z
    return _ZERO[key]


# ------------------------------------------------------------------------------------------------
@dataclass
class ConvW:
    """Packed weights of one conv / linear layer: fp16 [n_pad][k_pad], k = tap*cin_p + c."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    n: int            # logical output channels (GEGLU: 2*features)
    n_pad: int
    k_pad: int
    cin: int          # logical input channels
    cin_p: int        # padded input channels seen by the kernel (cin, or 8 in small mode)
    kt: int
    kh: int
    kw: int
    geglu: bool = False
    k_logical: int = 0          # multiply-adds per output element when the packed K axis carries structural zeros (0: kt*kh*kw*cin)

    @property
    def n_out(self):
        return self.n // 2 if self.geglu else self.n


class LnOperand:
    """What a LayerNorm-folding consumer needs of its input rows x (fp32): `raw` = fp16(x) [M][n] and `stat` = per row and
    128-column chunk (sum, sum of squares) of x, [n/128][M][2] fp32 — written by the linear that produced x (ln_produce)."""
    __slots__ = ("raw", "stat", "n", "version")

    def __init__(self, raw, stat, n):
        self.raw, self.stat, self.n, self.version = raw, stat, n, None


def ln_operand_of(x):
    op = getattr(x, "_uav_ln", None)
    if op is None or op.version != x._version or x.shape[-1] != op.n or op.raw.shape[0] != x.shape[0]:
        return None
    return op


def _round_up(x, m):
    return (x + m - 1) // m * m


def pack_conv(weight, bias=None, geglu=False, device=None, n_store_align=4):
    """Pack an nn.Conv2d (O,I,kh,kw) / nn.Conv3d (O,I,kt,kh,kw) / nn.Linear (O,I) weight.

    k index = ((dt*kh + dy)*kw + dx)*cin_p + c.  Inputs with cin % 64 != 0 use the "small" mode:
    channels padded to 8 (requires cin <= 8).  GEGLU: rows re-ordered in blocks of
    [32 value rows | 32 gate rows] (diffusers GEGLU chunk order: value first, gate second).
    """
    w = weight.detach()
    if w.dim() == 2:
        w = w[:, :, None, None, None]
    elif w.dim() == 4:
        w = w[:, :, None]
    assert w.dim() == 5
    o, i, kt, kh, kw = w.shape
    b = None if bias is None else bias.detach().float()
    if i % 64 == 0:
        cin_p = i
    elif i <= 8:
        cin_p = 8
    else:
        raise _lib.UavError(f"conv input channels {i} unsupported (need multiple of 64 or <= 8)")
    w = w.permute(0, 2, 3, 4, 1).float()                      # o, kt, kh, kw, c
    if cin_p != i:
        w = torch.nn.functional.pad(w, (0, cin_p - i))
    w = w.reshape(o, kt * kh * kw * cin_p)
    n = o
    if geglu:
        assert o % 64 == 0
        f = o // 2
        val = w[:f].reshape(f // 32, 32, -1)
        gate = w[f:].reshape(f // 32, 32, -1)
        w = torch.cat([val, gate], dim=1).reshape(o, -1)
        if b is not None:
            b = torch.cat([b[:f].reshape(f // 32, 32), b[f:].reshape(f // 32, 32)], dim=1).reshape(o)
    if n % n_store_align:                                      # e.g. VAE conv_out: 3 -> 4 (zero row)
        extra = n_store_align - n % n_store_align
        w = torch.nn.functional.pad(w, (0, 0, 0, extra))
        if b is not None:
            b = torch.nn.functional.pad(b, (0, extra))
        n += extra
    k = w.shape[1]
    n_pad, k_pad = _round_up(n, 128), _round_up(k, 64)
    wp = torch.zeros(n_pad, k_pad, dtype=HALF)
    wp[:n, :k] = w.to(HALF)
    bp = None
    if b is not None:
        bp = torch.zeros(n_pad, dtype=torch.float32)
        bp[:n] = b
    dev = device if device is not None else weight.device
    return ConvW(wp.to(dev), None if bp is None else bp.to(dev), n, n_pad, k_pad, i, cin_p, kt, kh, kw, geglu)


def pack_conv_with_shortcut(w_main, b_main, w_short, b_short, repeat_short=1, device=None):
    """`conv_shortcut(x) + conv2(h)` of a ResNet block (reference resnet.py:286-292) as ONE implicit GEMM: K = taps x (C + Cs),
    the 1x1 shortcut weights sit at the centre tap of the Cs extra input channels and every other entry of those channels is
    zero (uav_conv_params.a2_center_tap: the 256x256 kernel skips the zero k-steps).  repeat_short = 2: the shortcut operand
    arrives as [hi | lo] fp16 halves, its weights are repeated.  Bias = b_main + b_short (fp32)."""
    w2 = w_main.detach().float()
    ws = w_short.detach().float()
    if w2.dim() != 4 or ws.dim() not in (4, 5) or any(k != 1 for k in ws.shape[2:]) or not (w2.shape[2] & 1 and w2.shape[3] & 1):
        raise _lib.UavError("pack_conv_with_shortcut: (O,C,kh,kw) odd-sized main conv + 1x1 shortcut expected")
    o, c, kh, kw = w2.shape
    ws = ws.reshape(o, -1)
    cs = ws.shape[1] * repeat_short
    full = torch.zeros((o, c + cs, kh, kw), dtype=torch.float32, device=w2.device)
    full[:, :c] = w2
    full[:, c:, kh // 2, kw // 2] = torch.cat([ws] * repeat_short, dim=1)
    b = None
    if b_main is not None or b_short is not None:
        b = (0 if b_main is None else b_main.detach().float()) + (0 if b_short is None else b_short.detach().float())
    cw = pack_conv(full, b, device=device if device is not None else w_main.device)
    cw.k_logical = kh * kw * c + ws.shape[1]
    return cw


def upsample_phase_weights(weight):
    """Sub-pixel form of "nearest 2x upsampling, then 3x3 conv with padding 1" (reference resnet.py:144-158): output pixel
    (2y + py, 2x + px) only ever sees the 2x2 input pixels (y - 1 + py .., x - 1 + px ..), because two of the three taps
    of each axis fall on the same input pixel.  Returns the four (O, I, 2, 2) fp32 weights indexed [py][px]; phase
    (py, px) is a 2x2 conv with top / left padding (1 - py, 1 - px) and zero taps past the bottom / right edge.  Same
    result in exact arithmetic with 16 instead of 36 multiply-adds per output element; the tap sums are formed in fp32
    before the single fp16 rounding of the packed weights."""
    w = weight.detach().float()
    if w.dim() != 4 or tuple(w.shape[-2:]) != (3, 3):
        raise _lib.UavError("upsample_phase_weights expects (O, I, 3, 3)")
    rows = [torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], dim=2),        # py = 0: rows y-1, y
            torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], dim=2)]        # py = 1: rows y, y+1
    out = []
    for r in rows:                                                            # r: (O, I, 2, 3)
        out.append([torch.stack([r[..., 0], r[..., 1] + r[..., 2]], dim=3),   # px = 0: cols x-1, x
                    torch.stack([r[..., 0] + r[..., 1], r[..., 2]], dim=3)])  # px = 1: cols x, x+1
    return out


def conv_gemm(a1, wt: ConvW, *, n_img, t_len, hi, wi, stride=1, pad=None, upsample=False, a2=None,
              rowbias=None, rows_per_batch=0, residual=None, out_scale=1.0, out_f32=False, out=None, out_hw=None,
              persistent=False, act=None, gn_groups=None, out_map=None, a2_center=False, ln_produce=False, ln_consume=None,
              gn_shared=None, no_shortk=False, no_w4=False, out_hilo=False):
    """out[M][n_out] = scale*(conv(a1|a2, W) + bias + rowbias[m//rows_per_batch] + residual).

    out_hilo (with out_f32): return the fp32 result as its fp16 operand pair [M][2 n_out] = [fp16(v) | fp16(v - fp16(v))] — what
    `cast_hilo` makes of it, written by the epilogue itself where the launch allows (uav_conv_gemm_hilo_ok), else by that pass.

    no_shortk: keep a 1x1 launch out of the short-K kernel (UAV_CONV_NO_SHORTK: A/B measurements, bit-identity test).

    ln_produce: the fp32 output feeds a LayerNorm whose consumer folds it (`LnFold`): the launch also writes the fp16 rounding of
    the rows and per-row statistics partials, attached to the returned tensor as `_uav_ln` when the launch qualifies.
    ln_consume = (LnOperand, colsum, eps): a1 is `LnOperand.raw`; the epilogue applies rstd * (acc - mu * colsum) + bias.

    gn_groups: the output feeds a GroupNorm of that many groups — when the launch qualifies
    (`uav_conv_gemm_gn_chunk_rows`) the epilogue also reduces the statistics partials and the returned tensor carries
    them (`GnPartials`, attribute `_uav_gn`); `groupnorm_scale_shift` then skips its pass over the tensor.

    out_map = (w, sy, sx, off) with `out`: GEMM row m = Y*w + x lands in row Y*sy + x*sx + off of `out` (sub-pixel phases
    of the upsampling convs, `upsample_phase_weights`).
    gn_shared = (SharedGnPartials, cpi, stride, off) with out_map: this launch's statistics chunks go into a workspace shared
    with the other phase launches of the same output tensor (uav_conv_params.gn_chunk_*); the caller attaches the partials
    to `out` when every launch reported success (`SharedGnPartials.filled`)."""
    lib = _lib.load()
    _req(a1, HALF, "a1")
    c1 = a1.shape[-1]
    c2 = 0
    if a2 is not None:
        _req(a2, HALF, "a2")
        c2 = a2.shape[-1]
    if c1 + c2 != wt.cin_p:
        raise _lib.UavError(f"conv input channels {c1}+{c2} != packed {wt.cin_p}")
    if pad is None:
        pad = (wt.kt // 2, wt.kh // 2, wt.kw // 2)
    pt, ph, pw = pad
    if upsample:
        ho, wo = 2 * hi, 2 * wi
    else:
        ho = (hi + 2 * ph - wt.kh) // stride + 1
        wo = (wi + 2 * pw - wt.kw) // stride + 1
    if out_hw is not None:                 # asymmetric (right/bottom) zero padding: taps past the edge read zeros
        ho, wo = out_hw
    if a1.numel() != n_img * hi * wi * c1:
        raise _lib.UavError(f"a1 has {a1.numel()} elements, expected {n_img}*{hi}*{wi}*{c1}")
    a2_images = 0
    if a2 is not None and a2.numel() != n_img * hi * wi * c2:
        # a skip tensor of the CFG-shared UNet head: it exists once and serves both batch entries (read batch-broadcast)
        if n_img % 2 or a2.numel() * 2 != n_img * hi * wi * c2:
            raise _lib.UavError(f"a2 has {a2.numel()} elements, expected {n_img}*{hi}*{wi}*{c2} (or half of it)")
        a2_images = n_img // 2
    m = n_img * ho * wo
    n_out = wt.n_out
    if out_map is not None and (out is None or residual is not None):
        raise _lib.UavError("out_map needs a caller-supplied `out` and no residual")
    if out_hilo and (not out_f32 or out is not None or out_map is not None or ln_produce or wt.geglu):
        raise _lib.UavError("out_hilo is a storage form of a plain fp32 result (out_f32, library-allocated output)")
    if out is None:
        out = torch.empty((m, 2 * n_out), dtype=HALF, device=a1.device) if out_hilo else \
            torch.empty((m, n_out), dtype=torch.float32 if out_f32 else HALF, device=a1.device)
    flags = (_lib.CONV_GEGLU if wt.geglu else 0) | (_lib.CONV_OUT_F32 if out_f32 else 0) | (_lib.CONV_PERSISTENT if persistent else 0)
    if out_hilo:
        flags |= _lib.CONV_OUT_HILO
    if act is not None:
        flags |= {"gelu": _lib.CONV_GELU, "quick_gelu": _lib.CONV_QUICK_GELU}[act]
    if no_shortk:
        flags |= _lib.CONV_NO_SHORTK
    if no_w4:
        flags |= _lib.CONV_NO_W4
    p = _lib.ConvParams()
    p.a1 = _p(a1); p.a2 = _p(a2); p.c1 = c1; p.c2 = c2
    p.w = _p(wt.w); p.bias = _p(wt.bias)
    p.rowbias = _p(rowbias); p.rows_per_batch = rows_per_batch
    p.rowbias_stride = 0 if rowbias is None else rowbias.shape[-1]
    if rowbias is not None:
        _req(rowbias, torch.float32, "rowbias")
    p.residual = _p(residual); p.res_stride = 0 if residual is None else residual.shape[-1]
    if residual is not None:
        if residual.dtype == torch.float32:            # fp32 residual stream (VAE decoder)
            _req(residual, torch.float32, "residual")
            flags |= _lib.CONV_RES_F32
        else:
            _req(residual, HALF, "residual")
        if residual.numel() != m * residual.shape[-1]:
            raise _lib.UavError("residual rows != output rows")
    p.out = _p(out); p.out_stride = out.shape[-1]
    p.n_img = n_img; p.t_len = t_len; p.hi = hi; p.wi = wi; p.ho = ho; p.wo = wo
    p.kt = wt.kt; p.kh = wt.kh; p.kw = wt.kw; p.stride = stride
    p.pad_t = pt; p.pad_h = ph; p.pad_w = pw; p.upsample = 1 if upsample else 0
    p.n = wt.n; p.n_pad = wt.n_pad; p.k_pad = wt.k_pad
    p.out_scale = out_scale; p.flags = flags; p.zero_page = _p(zero_page(a1.device))
    p.a2_images = a2_images
    if a2_center:                    # a2 takes part in the centre tap only (shortcut conv folded into conv2, pack_conv_with_shortcut)
        if a2 is None:
            raise _lib.UavError("a2_center needs a second source")
        p.a2_center_tap = 1
    if out_map is not None:
        p.out_map_w, p.out_map_sy, p.out_map_sx, p.out_map_off = (int(v) for v in out_map)
        if (m // out_map[0] - 1) * out_map[1] + (out_map[0] - 1) * out_map[2] + out_map[3] >= out.shape[0]:
            raise _lib.UavError("out_map places rows past the end of `out`")
    hilo_pass = False
    if out_hilo and not lib.uav_conv_gemm_hilo_ok(C.byref(p)):
        # this launch cannot store the pair itself (small grid, tile tails, ...): fp32 result + the cast pass
        hilo_pass = True
        out = torch.empty((m, n_out), dtype=torch.float32, device=a1.device)
        p.out = _p(out); p.out_stride = n_out; p.flags = flags & ~_lib.CONV_OUT_HILO
    lnop = None
    if ln_produce and out_f32 and n_out % 128 == 0 and out.shape[-1] == n_out:
        lnop = LnOperand(torch.empty((m, n_out), dtype=HALF, device=a1.device),
                         torch.empty((n_out // 128, m, 2), dtype=torch.float32, device=a1.device), n_out)
        p.ln_raw_out = _p(lnop.raw); p.ln_stat_out = _p(lnop.stat)
        if not lib.uav_conv_gemm_ln_ok(C.byref(p)):
            lnop = None; p.ln_raw_out = None; p.ln_stat_out = None
    if ln_consume is not None:
        src, colsum, eps = ln_consume
        p.ln_stat_in = _p(src.stat); p.ln_colsum = _p(colsum); p.ln_chunks = src.stat.shape[0]; p.ln_n = src.n; p.ln_eps = float(eps)
        if not lib.uav_conv_gemm_ln_ok(C.byref(p)):
            raise _lib.UavError("conv_gemm: this launch cannot fold the LayerNorm (ask ln_fold_ok first)")
    gn = None
    if gn_shared is not None and FUSE_GN_STATS and lnop is None:
        sh, cpi, cstride, coff = gn_shared
        p.gn_groups = int(sh.groups)
        p.gn_chunk_cpi, p.gn_chunk_stride, p.gn_chunk_off, p.gn_chunks_total = int(cpi), int(cstride), int(coff), int(sh.ws.shape[-1])
        if lib.uav_conv_gemm_gn_chunk_rows(C.byref(p)) == sh.rows and (m // sh.rows) % cpi == 0:
            p.gn_partials = _p(sh.ws)
            sh.filled += 1
        else:
            p.gn_groups = 0; p.gn_chunk_cpi = 0
    elif gn_groups and FUSE_GN_STATS and out_map is None and lnop is None and not out_hilo:
        p.gn_groups = int(gn_groups)
        rows = lib.uav_conv_gemm_gn_chunk_rows(C.byref(p))
        if rows > 0:
            gn = GnPartials(torch.empty((2, int(gn_groups), m // rows), dtype=torch.float32, device=a1.device), rows,
                            int(gn_groups), n_out)
            p.gn_partials = _p(gn.ws)
        else:
            p.gn_groups = 0
    ev = PROFILER.begin("conv_gemm")
    _lib.check(lib.uav_conv_gemm_f16(C.byref(p), _stream()), "uav_conv_gemm_f16")
    if gn is not None:
        _gn_attach(out, gn)
    elif getattr(out, "_uav_gn", None) is not None:      # caller-supplied buffer rewritten without statistics
        out._uav_gn = None
    if lnop is not None:
        lnop.version = out._version
        out._uav_ln = lnop
    elif getattr(out, "_uav_ln", None) is not None:      # caller-supplied buffer rewritten without the LayerNorm operand copy
        out._uav_ln = None
    # algorithmic work: 2*M*N*K over the LOGICAL taps x input channels (no channel / tile padding counted); temporal taps
    # that fall outside the clip (zero padding of a (k,1,1) / 3x3x3 conv at the clip ends, which the kernel skips) are
    # not counted either
    tfrac = 1.0
    if wt.k_logical:
        tfrac = wt.k_logical / float(wt.kt * wt.kh * wt.kw * wt.cin)
    if ev is not None and wt.kt > 1:
        valid = sum(1 for t_ in range(t_len) for dt in range(wt.kt) if 0 <= t_ + dt - pt < t_len)
        tfrac = valid / float(wt.kt * t_len)
    PROFILER.end(ev, "conv_gemm" if not PROFILER.detail else
                 f"conv_gemm cin={wt.cin} n={wt.n} k={wt.kt}x{wt.kh}x{wt.kw} M={m}{' geglu' if wt.geglu else ''}{' up' if upsample else ''}{' s2' if stride == 2 else ''}",
                 2.0 * m * wt.n * wt.kt * wt.kh * wt.kw * wt.cin * tfrac,
                 # algorithmic bytes: every operand once — X and W fp16, the result and the residual in their stored type
                 2.0 * n_img * hi * wi * wt.cin + 2.0 * wt.n * wt.kt * wt.kh * wt.kw * wt.cin + (4.0 if out_f32 else 2.0) * m * n_out
                 + (0.0 if residual is None else (4.0 if residual.dtype == torch.float32 else 2.0) * m * n_out))
    if hilo_pass:
        return cast_hilo(out)
    return out


@functools.lru_cache(maxsize=256)
def _factor_rows(m):
    """m = n_img * hi with hi < 65536 (the kernel packs pixel coordinates in 16 bits).  Memoised: the divisor search is
    a Python loop (~0.6 ms for the UNet's token counts) and the UNet issues ~160 linears per forward."""
    if m < 65536:
        return 1, m
    for hi in range(min(m, 65535), 0, -1):
        if m % hi == 0:
            return m // hi, hi
    raise _lib.UavError("unreachable")


def linear(x, wt: ConvW, *, residual=None, out_scale=1.0, rowbias=None, rows_per_batch=0, out_f32=False, act=None,
           gn_groups=None, ln_produce=False, ln_consume=None, no_shortk=False, no_w4=False, out_hilo=False):
    """nn.Linear over token rows x[M][K] (a 1x1 'conv': every row is one pixel)."""
    n_img, hi = _factor_rows(x.shape[0])
    return conv_gemm(x, wt, n_img=n_img, t_len=1, hi=hi, wi=1, residual=residual, out_scale=out_scale,
                     rowbias=rowbias, rows_per_batch=rows_per_batch, out_f32=out_f32, act=act, gn_groups=gn_groups,
                     ln_produce=ln_produce, ln_consume=ln_consume, no_shortk=no_shortk, no_w4=no_w4, out_hilo=out_hilo)


def ln_fold_ok(m, k, wt: ConvW):
    """Can a linear over m rows of k channels with packed weights `wt` fold a LayerNorm (uav_conv_gemm_ln_ok)?  Mirrors the C
    check without building a launch: 256x256 kernel (enough tiles), full wave tiles."""
    n_img, hi = _factor_rows(m)
    p = _lib.ConvParams()
    p.c1 = k; p.c2 = 0; p.n_img = n_img; p.t_len = 1; p.hi = hi; p.wi = 1; p.ho = hi; p.wo = 1; p.kt = p.kh = p.kw = 1; p.stride = 1
    p.n = wt.n; p.n_pad = wt.n_pad; p.k_pad = wt.k_pad; p.out_stride = wt.n_out
    p.flags = _lib.CONV_GEGLU if wt.geglu else 0
    p.bias = 1; p.ln_stat_in = 1; p.ln_colsum = 1; p.ln_chunks = k // 128; p.ln_n = k     # non-NULL markers: host-only query
    return bool(_lib.load().uav_conv_gemm_ln_ok(C.byref(p)))


# ------------------------------------------------------------------------------------------------
# GroupNorm statistics produced by the conv that wrote the tensor (uav_conv_params.gn_partials).  The partials ride on
# the tensor OBJECT the conv returned: a view, a copy or an in-place update of it (`_version` moves) does not carry them,
# and the GroupNorm then runs its own statistics pass.
FUSE_GN_STATS = os.environ.get("UAV_FUSE_GN_STATS", "1") != "0"
GN_TWO_SOURCE = os.environ.get("UAV_GN_TWO_SOURCE", "1") != "0"      # concatenated inputs: statistics from the two producers' partials


class GnPartials:
    __slots__ = ("ws", "rows", "groups", "c", "version")

    def __init__(self, ws, rows, groups, c):
        self.ws, self.rows, self.groups, self.c, self.version = ws, rows, groups, c, None


class SharedGnPartials(GnPartials):
    """One statistics workspace filled by several launches that write interleaved rows of one tensor (the four sub-pixel phase
    convs of an up-sampler): `filled` counts the launches that did write their chunks."""
    __slots__ = ("filled",)

    def __init__(self, rows_out, groups, c, device, chunk_rows=64):
        super().__init__(torch.empty((2, int(groups), rows_out // chunk_rows), dtype=torch.float32, device=device), chunk_rows,
                         int(groups), c)
        self.filled = 0


def _gn_attach(t, gn):
    gn.version = t._version
    t._uav_gn = gn


def _gn_partials_of(x, groups, c, rows_per_inst):
    gn = getattr(x, "_uav_gn", None)
    if gn is None or gn.version != x._version or gn.groups != groups or gn.c != c or x.shape[-1] != c:
        return None
    if rows_per_inst % gn.rows:
        return None
    if gn.ws.shape[-1] * gn.rows != x.shape[0]:      # partials of another row count (the tensor was re-interpreted): stats pass
        return None
    return gn


def _gn_partials_any(x):
    """Valid statistics partials riding on `x`, whatever group count their producer assumed."""
    gn = getattr(x, "_uav_gn", None)
    if gn is None or gn.version != x._version or x.shape[-1] != gn.c or gn.ws.shape[-1] * gn.rows != x.shape[0]:
        return None
    return gn


def _gn_two_source(x1, x2, groups, n_inst, rows_per_inst):
    """(gn1, gn2, n_inst2) when the statistics of the concatenated input [x1 | x2] can be had from the two producers'
    partials (uav_groupnorm_finalize_partials2), else None."""
    g1, g2 = _gn_partials_any(x1), _gn_partials_any(x2)
    if g1 is None or g2 is None or g1.rows != g2.rows or rows_per_inst % g1.rows:
        return None
    c1, c2 = x1.shape[-1], x2.shape[-1]
    c = c1 + c2
    if c % groups:
        return None
    cpg = c // groups
    if c1 % cpg or cpg % (c1 // g1.groups) or cpg % (c2 // g2.groups):
        return None
    rows = n_inst * rows_per_inst
    if x1.shape[0] != rows or x2.shape[0] not in (rows, rows // 2) or (x2.shape[0] != rows and (n_inst % 2 or rows % 2)):
        return None
    n2 = n_inst if x2.shape[0] == rows else n_inst // 2
    return g1, g2, n2


def duplicate_rows(t):
    """[t; t] along the row axis (classifier-free guidance: one evaluation of the text-free head serves both batch
    entries).  Statistics partials riding on `t` are duplicated with it, so the GroupNorm that follows sees exactly what
    it would have seen after evaluating the head on the duplicated batch."""
    out = torch.cat([t, t])
    gn = getattr(t, "_uav_gn", None)
    if gn is not None and gn.version == t._version:
        _gn_attach(out, GnPartials(torch.cat([gn.ws, gn.ws], dim=-1), gn.rows, gn.groups, gn.c))
    return out


def _gn_x2_rows(x1, x2, rows):
    """0, or the row count of a second source that exists once for both batch entries (read batch-broadcast)."""
    if x2 is None or x2.shape[0] == rows:
        return 0
    if x2.shape[0] * 2 != rows:
        raise _lib.UavError(f"groupnorm: second source has {x2.shape[0]} rows, expected {rows} (or half of it)")
    return x2.shape[0]


def _gn_dtype(x1, x2):
    """GroupNorm inputs are fp16 rows, or fp32 rows (fp32 residual stream of the VAE decoder); both sources alike."""
    if x1.dtype not in (HALF, torch.float32):
        raise _lib.UavError(f"groupnorm input must be fp16 or fp32, got {x1.dtype}")
    _req(x1, x1.dtype, "x1")
    if x2 is not None:
        _req(x2, x1.dtype, "x2")
    return 1 if x1.dtype == torch.float32 else 0


def groupnorm_scale_shift(x1, gamma, beta, *, n_inst, rows_per_inst, groups, eps, x2=None, c_real=None):
    lib = _lib.load()
    xf32 = _gn_dtype(x1, x2)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    c = c1 + c2
    if c_real is None:
        c_real = c
    scale = torch.empty((n_inst, c), dtype=torch.float32, device=x1.device)
    shift = torch.empty_like(scale)
    gn = _gn_partials_of(x1, groups, c, rows_per_inst) if (x2 is None and c_real == c) else None
    if gn is not None and gn.ws.shape[-1] * gn.rows != n_inst * rows_per_inst:
        gn = None                                    # instance split does not cover the producer's rows: statistics pass
    two = _gn_two_source(x1, x2, groups, n_inst, rows_per_inst) if (x2 is not None and c_real == c and FUSE_GN_STATS and GN_TWO_SOURCE) else None
    if two is not None:
        g1, g2, n2 = two
        ev = PROFILER.begin("groupnorm_stats")
        rc = lib.uav_groupnorm_finalize_partials2(_p(g1.ws), g1.ws.shape[-1], c1, g1.groups, n_inst, _p(g2.ws), g2.ws.shape[-1], c2,
                                                  g2.groups, n2, g1.rows, n_inst, rows_per_inst, groups, eps, _p(gamma), _p(beta),
                                                  _p(scale), _p(shift), _stream())
        _lib.check(rc, "uav_groupnorm_finalize_partials2")
        PROFILER.end(ev, "groupnorm_finalize_fused", 0.0, 4.0 * (g1.ws.numel() + g2.ws.numel()))
        return scale, shift
    if gn is not None:
        chunks_total = gn.ws.shape[-1]
        ev = PROFILER.begin("groupnorm_stats")
        rc = lib.uav_groupnorm_finalize_partials(_p(gn.ws), chunks_total, gn.rows, c, n_inst, rows_per_inst, groups, eps,
                                                 _p(gamma), _p(beta), _p(scale), _p(shift), _stream())
        _lib.check(rc, "uav_groupnorm_finalize_partials")
        PROFILER.end(ev, "groupnorm_finalize_fused", 0.0, 4.0 * gn.ws.numel())
        return scale, shift
    ws_bytes = lib.uav_groupnorm_workspace_bytes(n_inst, c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x1.device)
    ev = PROFILER.begin("groupnorm_stats")
    rc = lib.uav_groupnorm_scale_shift(_p(x1), _p(x2), xf32, c1, c2, _gn_x2_rows(x1, x2, n_inst * rows_per_inst), c_real,
                                       n_inst, rows_per_inst, groups, eps,
                                       _p(gamma), _p(beta), _p(scale), _p(shift), _p(ws), ws_bytes, _stream())
    _lib.check(rc, "uav_groupnorm_scale_shift")
    PROFILER.end(ev, "groupnorm_stats", 0.0, (4.0 if xf32 else 2.0) * n_inst * rows_per_inst * c)
    return scale, shift


def groupnorm_apply(x1, scale, shift, *, n_inst, rows_per_inst, silu, x2=None, want_raw=False):
    """want_raw: also return the un-normalised [x1|x2] rows as fp16 (extra write in the same pass): True -> rows of c values
    fp16(x); "hilo" -> rows of 2c values [hi | lo], hi = fp16(x), lo = fp16(x - hi) (x to ~22 bits as two MFMA operands)."""
    lib = _lib.load()
    xf32 = _gn_dtype(x1, x2)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    y = torch.empty((n_inst * rows_per_inst, c1 + c2), dtype=HALF, device=x1.device)
    hilo = want_raw == "hilo"
    raw = torch.empty((y.shape[0], (2 if hilo else 1) * (c1 + c2)), dtype=HALF, device=x1.device) if want_raw else None
    ev = PROFILER.begin("groupnorm_apply")
    rc = lib.uav_groupnorm_apply(_p(x1), _p(x2), xf32, c1, c2, _gn_x2_rows(x1, x2, n_inst * rows_per_inst), n_inst,
                                 rows_per_inst, _p(scale), _p(shift), 1 if silu else 0, _p(y), _p(raw), int(hilo), _stream())
    _lib.check(rc, "uav_groupnorm_apply")
    PROFILER.end(ev, "groupnorm_apply", 0.0, ((6.0 if xf32 else 4.0) + (4.0 if hilo else 2.0 if want_raw else 0.0)) * n_inst * rows_per_inst * (c1 + c2))
    return (y, raw) if want_raw else y


def groupnorm(x1, gamma, beta, *, n_inst, rows_per_inst, groups, eps, silu, x2=None, c_real=None, want_raw=False):
    """GroupNorm(+SiLU) over `n_inst` instances of `rows_per_inst` channels-last rows.  want_raw: returns (y, raw16), see
    groupnorm_apply."""
    sc, sh = groupnorm_scale_shift(x1, gamma, beta, n_inst=n_inst, rows_per_inst=rows_per_inst, groups=groups,
                                   eps=eps, x2=x2, c_real=c_real)
    return groupnorm_apply(x1, sc, sh, n_inst=n_inst, rows_per_inst=rows_per_inst, silu=silu, x2=x2, want_raw=want_raw)


def layernorm(x, gamma, beta, eps=1e-5):
    """fp16 LayerNorm output (an MFMA operand) of fp16 rows, or of fp32 rows (fp32 residual stream)."""
    lib = _lib.load()
    f32 = x.dtype == torch.float32
    _req(x, torch.float32 if f32 else HALF, "x")
    y = torch.empty(x.shape, dtype=HALF, device=x.device)
    rows, c = x.numel() // x.shape[-1], x.shape[-1]
    ev = PROFILER.begin("layernorm")
    fn = lib.uav_layernorm_f32in if f32 else lib.uav_layernorm_f16
    _lib.check(fn(_p(x), _p(y), _p(gamma), _p(beta), rows, c, eps, _stream()), "uav_layernorm")
    PROFILER.end(ev, "layernorm", 0.0, (6.0 if f32 else 4.0) * rows * c)
    return y


# ------------------------------------------------------------------------------------------------
# d = 512 single-head attention (VAE) on the ring kernel of csrc/attn512x.hip; UAV_ATTN512X=0 keeps attn512w_kernel (attention.hip)
ATTN512X = os.environ.get("UAV_ATTN512X", "1") != "0"


def attention(q, k, v, *, bq, lq, lk, heads, head_dim, q_per_kv=1, scale=None,
              q_stride=None, k_stride=None, v_stride=None, causal=False):
    """softmax(scale*QK^T)V.  q/k/v are fp16 views whose row strides (in elements) may exceed
    heads*head_dim (slices of a fused projection)."""
    lib = _lib.load()
    c = heads * head_dim
    q_stride = q_stride or q.stride(-2)
    k_stride = k_stride or k.stride(-2)
    v_stride = v_stride or v.stride(-2)
    out = torch.empty((bq * lq, c), dtype=HALF, device=q.device)
    if scale is None:
        scale = head_dim ** -0.5
    if ATTN512X and heads == 1 and head_dim == 512 and q_per_kv == 1 and not causal and lk >= 1024:
        # the VAE mid-block attention (d = 512, L = H W): K / V^T re-packed once per call into the MFMA-fragment streams the ring kernel walks
        # (csrc/attn512x.hip); the 0.2 GB per frame of workspace comes from the caching allocator and is released on return
        nb = lib.uav_attention512_pack_bytes(bq, lk)
        kp = torch.empty(nb, dtype=torch.uint8, device=q.device); vp = torch.empty(nb, dtype=torch.uint8, device=q.device)
        ev = PROFILER.begin("attention")
        _lib.check(lib.uav_attention512_pack_kv(_p(k), k_stride, _p(v), v_stride, bq, lk, _p(kp), _p(vp), _stream()), "uav_attention512_pack_kv")
        _lib.check(lib.uav_attention512_packed_f16(_p(q), q_stride, _p(kp), _p(vp), _p(out), c, bq, lq, lk, scale, _stream()),
                   "uav_attention512_packed_f16")
        PROFILER.end(ev, "attention_d512", 4.0 * bq * lq * lk * head_dim, 2.0 * (2 * bq * lq * c + 6 * bq * lk * c))
        return out
    ev = PROFILER.begin("attention")
    rc = lib.uav_attention_f16(_p(q), q_stride, _p(k), k_stride, _p(v), v_stride, _p(out), c, bq, lq, lk, q_per_kv,
                               heads, head_dim, scale, int(causal), _p(zero_page(q.device)), _stream())
    _lib.check(rc, "uav_attention_f16")
    PROFILER.end(ev, f"attention_d{head_dim}", 4.0 * bq * heads * lq * lk * head_dim, 2.0 * (2 * bq * lq * c + 2 * (bq // q_per_kv) * lk * c))
    return out


# ------------------------------------------------------------------------------------------------
# Fused text cross-attention sub-layer (csrc/xattn_fused.hip): LayerNorm -> to_q -> softmax(Q K^T) V -> to_out -> + residual
XATTN_C, XATTN_HEADS, XATTN_D, XATTN_MAX_KEYS, XATTN_TILE = 512, 8, 64, 96, 128


def pack_xattn_weight(weight, kind, device=None):
    """nn.Linear weight [512 out][512 in] of to_q (kind 'q') / to_out (kind 'out') -> the A-fragment stream the fused sub-layer
    kernel walks: 1-KiB fragments of 32 rows x 16 k, lane (hi, l32) holds row l32 and k = 16 ks + 8 e2 + 4 hi + e1 in slots
    e = 4 e2 + e1.  'q': head h, fragment g = (k-step g >> 1, channel tile g & 1) of rows 64 h + 32 tile + l32;
    'out': head h, fragment g = (channel tile 2 (g >> 3) + (g & 1), k-step (g >> 1) & 3) of rows 32 tile + l32, k inside head h."""
    w = weight.detach().float()
    if tuple(w.shape) != (XATTN_C, XATTN_C):
        raise _lib.UavError(f"fused cross-attention weights must be {XATTN_C}x{XATTN_C}, got {tuple(w.shape)}")
    if kind == "q":
        # rows = (h, mt, l32); k = (ks, e2, hi, e1)  ->  [h][ks][mt][hi][l32][e2][e1]
        p = w.reshape(8, 2, 32, 32, 2, 2, 4).permute(0, 3, 1, 5, 2, 4, 6)
    elif kind == "out":
        # rows = (npair, par, l32) with tile = 2 npair + par; k = (h, ks, e2, hi, e1)  ->  [h][npair][ks][par][hi][l32][e2][e1]
        p = w.reshape(8, 2, 32, 8, 4, 2, 2, 4).permute(3, 0, 4, 1, 6, 2, 5, 7)
    else:
        raise ValueError(kind)
    out = p.contiguous().to(HALF).reshape(-1)
    assert out.numel() == XATTN_C * XATTN_C
    return out.to(device if device is not None else weight.device)


FF_INNER = 2048


def pack_ff_weights(w_up, w_down, device=None):
    """GEGLU proj weight [4096][512] (value rows | gate rows) and the down Linear weight [512][2048] -> the fragment stream of the fused
    feed-forward kernel (csrc/xattn_fused.hip ff_sublayer_kernel): per slice c of 32 hidden channels 96 KiB — 64 KiB of 'q' fragments
    (k-step, channel tile: tile 0 = value rows 32 c .., tile 1 = gate rows 2048 + 32 c ..) and 32 KiB of 'out' fragments of columns
    32 c .. of w_down (fragment g = (channel tile 2 (g >> 2) + (g & 1), k-step (g >> 1) & 1))."""
    wu, wd = w_up.detach().float(), w_down.detach().float()
    if tuple(wu.shape) != (2 * FF_INNER, XATTN_C) or tuple(wd.shape) != (XATTN_C, FF_INNER):
        raise _lib.UavError(f"fused feed-forward weights must be {2 * FF_INNER}x{XATTN_C} / {XATTN_C}x{FF_INNER}, got {tuple(wu.shape)} / {tuple(wd.shape)}")
    ns = FF_INNER // 32
    # rows = (vg, c, l32); k = (ks, e2, hi, e1)  ->  [c][ks][vg][hi][l32][e2][e1]
    u = wu.reshape(2, ns, 32, 32, 2, 2, 4).permute(1, 3, 0, 5, 2, 4, 6).reshape(ns, -1)
    # rows = (npair, par, l32); k = (c, ks, e2, hi, e1)  ->  [c][npair][ks][par][hi][l32][e2][e1]
    d = wd.reshape(8, 2, 32, ns, 2, 2, 2, 4).permute(3, 0, 4, 1, 6, 2, 5, 7).reshape(ns, -1)
    out = torch.cat([u, d], dim=1).contiguous().to(HALF).reshape(-1)
    assert out.numel() == 3 * FF_INNER * XATTN_C
    return out.to(device if device is not None else w_up.device)


def ff_ok(x, *, inner):
    """Shapes the fused feed-forward kernel takes."""
    return x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == XATTN_C and inner == FF_INNER and x.shape[0] % XATTN_TILE == 0


def ff_sublayer(x, gamma, beta, eps, w_packed, up_bias, down_bias, *, out_f32=True, out_hilo=False):
    """x + down(GEGLU(up(LayerNorm(x)))) on fp32 stream rows [M][512] in one launch.  Returns the fp32 rows, or (out_hilo) their fp16
    hi | lo operand pair [M][1024] (cast_hilo's layout), or (both flags) the tuple (rows, pair)."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    m = x.shape[0]
    if not (out_f32 or out_hilo):
        raise _lib.UavError("ff_sublayer: nothing to write")
    y = torch.empty_like(x) if out_f32 else None
    yh = torch.empty((m, 2 * XATTN_C), dtype=HALF, device=x.device) if out_hilo else None
    q = _lib.FfParams()
    q.ln_gamma, q.ln_beta, q.ln_eps = _p(gamma), _p(beta), float(eps)
    q.w_packed, q.up_bias, q.down_bias = _p(w_packed), _p(up_bias), _p(down_bias)
    ev = PROFILER.begin("ff_sublayer")
    rc = lib.uav_ff_sublayer_f32(_p(x), _p(y) if y is not None else None, _p(yh) if yh is not None else None, C.byref(q), m, XATTN_C, FF_INNER,
                                 _stream())
    _lib.check(rc, "uav_ff_sublayer_f32")
    PROFILER.end(ev, "ff_sublayer" if not PROFILER.detail else f"ff_sublayer M={m}", 2.0 * m * XATTN_C * 3 * FF_INNER,
                 4.0 * m * XATTN_C * (1 + (1 if out_f32 else 0) + (1 if out_hilo else 0)))
    if out_f32 and out_hilo:
        return y, yh
    return y if out_f32 else yh


def xattn_pack_kv(k, v, *, n_batch, lk, k_stride=None, v_stride=None):
    """Text K / V rows (fp16 views of the fused k|v projection) -> fragment stream [n_batch][8][32 KiB]."""
    lib = _lib.load()
    out = torch.empty((n_batch, XATTN_HEADS, 32 * 64 * 8), dtype=HALF, device=k.device)
    rc = lib.uav_xattn_pack_kv(_p(k), k_stride or k.stride(-2), _p(v), v_stride or v.stride(-2), n_batch, lk, XATTN_HEADS, XATTN_D,
                               _p(out), _stream())
    _lib.check(rc, "uav_xattn_pack_kv")
    return out


def xattn_ok(x, *, heads, head_dim, lk, rows_per_kv):
    """Shapes the fused sub-layer kernel takes (else the caller keeps the four-launch chain)."""
    return (x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == XATTN_C and heads == XATTN_HEADS and head_dim == XATTN_D
            and 0 < lk <= XATTN_MAX_KEYS and rows_per_kv % XATTN_TILE == 0 and x.shape[0] % rows_per_kv == 0)


def xattn_sublayers(x, subs, *, rows_per_kv, lk, scale, out=None):
    """One or two consecutive fused cross-attention sub-layers on fp32 stream rows x [M][512] in ONE launch:
    x <- x + to_out(attention(to_q(LayerNorm(x)), K, V)) + bias, for each entry of `subs` = (gamma, beta, eps, wq_packed, kv_packed,
    wo_packed, out_bias) in order (two entries: attn1 with only_cross_attention and attn2 of one BasicTransformerBlock — the rows
    between the two stay in the accumulators)."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    if not 1 <= len(subs) <= 2:
        raise _lib.UavError("xattn_sublayers takes one or two sub-layers")
    m = x.shape[0]
    y = torch.empty_like(x) if out is None else out
    arr = (_lib.XattnParams * len(subs))()
    for i, (gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias) in enumerate(subs):
        arr[i].ln_gamma, arr[i].ln_beta, arr[i].ln_eps = _p(gamma), _p(beta), float(eps)
        arr[i].wq_packed, arr[i].kv_packed, arr[i].wo_packed, arr[i].out_bias = _p(wq_packed), _p(kv_packed), _p(wo_packed), _p(out_bias)
    ev = PROFILER.begin("xattn_sublayer")
    rc = lib.uav_xattn_sublayers_f32(_p(x), _p(y), C.cast(arr, C.c_void_p), len(subs), m, rows_per_kv, lk, XATTN_C, XATTN_HEADS, scale, _stream())
    _lib.check(rc, "uav_xattn_sublayers_f32")
    n = len(subs)
    PROFILER.end(ev, "xattn_sublayer" if not PROFILER.detail else f"xattn_sublayer x{n} M={m}", n * (2.0 * m * (2 * XATTN_C * XATTN_C) + 4.0 * m * lk * XATTN_C),
                 8.0 * m * XATTN_C)
    return y


def xattn_sublayer(x, gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias, *, rows_per_kv, lk, scale, out=None):
    """A single fused sub-layer (see xattn_sublayers)."""
    return xattn_sublayers(x, [(gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias)], rows_per_kv=rows_per_kv, lk=lk, scale=scale, out=out)


def tattn_ok(x, *, heads, head_dim, t_len, hw, rot_dim):
    """Shapes the fused temporal sub-layer kernel takes (csrc/xattn_fused.hip tattn_sublayer_kernel)."""
    return (x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == XATTN_C and heads == XATTN_HEADS and head_dim == XATTN_D
            and t_len == 8 and rot_dim == 32 and hw % 16 == 0 and x.shape[0] % (t_len * hw) == 0)


def tattn_sublayer(x, gamma, beta, eps, wq_packed, wk_packed, wv_packed, wo_packed, out_bias, rel_bias, rope_cos, rope_sin, *,
                   n_batch, t_len, hw, rot_dim, scale, out=None, next_ln=None):
    """x + to_out(temporal_attention(to_q | to_k | to_v (LayerNorm(x)))) + bias on fp32 stream rows [B*T*hw][512] in one launch.
    next_ln = (gamma, beta, eps): also write fp16 LayerNorm(y) of the result rows (attached to y, see NextLn)."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    y = torch.empty_like(x) if out is None else out
    q = _tattn_params((gamma, beta, eps, wq_packed, wk_packed, wv_packed, wo_packed, out_bias, rel_bias, rope_cos, rope_sin, rot_dim))
    nl = _attach_next_ln(y, q, next_ln)
    m = x.shape[0]
    ev = PROFILER.begin("tattn_sublayer")
    rc = lib.uav_tattn_sublayer_f32(_p(x), _p(y), C.byref(q), n_batch, t_len, hw, XATTN_C, XATTN_HEADS, scale, _stream())
    _lib.check(rc, "uav_tattn_sublayer_f32")
    PROFILER.end(ev, "tattn_sublayer" if not PROFILER.detail else f"tattn_sublayer M={m}", 2.0 * m * (4 * XATTN_C * XATTN_C) + 4.0 * m * t_len * XATTN_C,
                 8.0 * m * XATTN_C)
    if nl is not None:
        nl.version = y._version
        y._uav_next_ln = nl
    elif getattr(y, "_uav_next_ln", None) is not None:
        y._uav_next_ln = None
    return y


class NextLn:
    """fp16 rows n = LayerNorm(y) that the kernel which produced y wrote beside it (`next_ln` of tattn_sublayer / block_attn_sublayers):
    attached to y as `_uav_next_ln`; engine.ln_linear uses it instead of a LayerNorm launch when it was made with the same parameters."""
    __slots__ = ("rows", "gamma", "beta", "eps", "version")

    def __init__(self, rows, gamma, beta, eps):
        self.rows, self.gamma, self.beta, self.eps, self.version = rows, gamma, beta, float(eps), None


def next_ln_of(x, gamma, beta, eps):
    nl = getattr(x, "_uav_next_ln", None)
    if nl is None or nl.version != x._version or nl.gamma is not gamma or nl.beta is not beta or nl.eps != float(eps) or nl.rows.shape != x.shape:
        return None
    return nl.rows


def _attach_next_ln(y, q, next_ln):
    """Fill the next-LayerNorm fields of TattnParams `q` for output rows y; returns the NextLn to attach after the launch."""
    if next_ln is None:
        return None
    gamma, beta, eps = next_ln
    nl = NextLn(torch.empty(y.shape, dtype=HALF, device=y.device), gamma, beta, eps)
    q.next_ln_out, q.next_ln_gamma, q.next_ln_beta, q.next_ln_eps = _p(nl.rows), _p(gamma), _p(beta), float(eps)
    return nl


def _tattn_params(t):
    gamma, beta, eps, wq_packed, wk_packed, wv_packed, wo_packed, out_bias, rel_bias, rope_cos, rope_sin, rot_dim = t
    q = _lib.TattnParams()
    q.ln_gamma, q.ln_beta, q.ln_eps = _p(gamma), _p(beta), float(eps)
    q.wq_packed, q.wk_packed, q.wv_packed, q.wo_packed, q.out_bias = _p(wq_packed), _p(wk_packed), _p(wv_packed), _p(wo_packed), _p(out_bias)
    q.rel_bias, q.rope_cos, q.rope_sin, q.rot_dim = _p(rel_bias), _p(rope_cos), _p(rope_sin), int(rot_dim)
    return q


def block_attn_sublayers(x, cross, temporal, *, n_batch, t_len, hw, lk, cross_scale, temporal_scale, out=None, next_ln=None):
    """attn1 -> attn2 -> attn_temporal of one BasicTransformerBlock (only_cross_attention) on fp32 stream rows [B*T*hw][512] in ONE
    launch: `cross` = two (gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias) tuples as for xattn_sublayers, `temporal` =
    (gamma, beta, eps, wq, wk, wv, wo packed, out_bias, rel_bias, rope_cos, rope_sin, rot_dim) as for tattn_sublayer."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    if len(cross) != 2:
        raise _lib.UavError("block_attn_sublayers takes the two cross-attention sub-layers of a block")
    y = torch.empty_like(x) if out is None else out
    arr = (_lib.XattnParams * 2)()
    for i, (gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias) in enumerate(cross):
        arr[i].ln_gamma, arr[i].ln_beta, arr[i].ln_eps = _p(gamma), _p(beta), float(eps)
        arr[i].wq_packed, arr[i].kv_packed, arr[i].wo_packed, arr[i].out_bias = _p(wq_packed), _p(kv_packed), _p(wo_packed), _p(out_bias)
    q = _tattn_params(temporal)
    nl = _attach_next_ln(y, q, next_ln)         # next_ln = (gamma, beta, eps) of the LayerNorm behind the three sub-layers (the block's norm3)
    m = x.shape[0]
    ev = PROFILER.begin("block_attn_sublayers")
    rc = lib.uav_block_attn_sublayers_f32(_p(x), _p(y), C.cast(arr, C.c_void_p), lk, cross_scale, C.byref(q), n_batch, t_len, hw, XATTN_C, XATTN_HEADS,
                                          temporal_scale, _stream())
    _lib.check(rc, "uav_block_attn_sublayers_f32")
    if nl is not None:
        nl.version = y._version
        y._uav_next_ln = nl
    elif getattr(y, "_uav_next_ln", None) is not None:
        y._uav_next_ln = None
    PROFILER.end(ev, "block_attn_sublayers" if not PROFILER.detail else f"block_attn_sublayers M={m}",
                 2.0 * m * (8 * XATTN_C * XATTN_C) + 2 * 4.0 * m * lk * XATTN_C + 4.0 * m * t_len * XATTN_C, 8.0 * m * XATTN_C)
    return y


def block_sublayers(x, cross, temporal, ff, *, n_batch, t_len, hw, lk, cross_scale, temporal_scale, out_f32=True, out_hilo=False, proj_in=None):
    """The whole BasicTransformerBlock (attn1 -> attn2 -> attn_temporal -> ff) on fp32 stream rows [B*T*hw][512] in ONE launch: `cross`,
    `temporal` as for block_attn_sublayers, `ff` = (gamma, beta, eps, w_packed, up_bias, down_bias) as for ff_sublayer.  Returns the fp32
    rows, or (out_hilo) their fp16 hi | lo pair, or (both) the tuple.  proj_in = (gn_scale, gn_shift, w_packed, bias): x is the input of
    the Transformer3DModel's GroupNorm and the launch starts with GroupNorm apply -> proj_in (scale / shift: groupnorm_scale_shift's per-frame
    rows; w_packed: pack_xattn_weight(proj_in.weight, 'out'))."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    if len(cross) != 2:
        raise _lib.UavError("block_sublayers takes the two cross-attention sub-layers of a block")
    if not (out_f32 or out_hilo):
        raise _lib.UavError("block_sublayers: nothing to write")
    m = x.shape[0]
    y = torch.empty_like(x) if out_f32 else None
    yh = torch.empty((m, 2 * XATTN_C), dtype=HALF, device=x.device) if out_hilo else None
    arr = (_lib.XattnParams * 2)()
    for i, (gamma, beta, eps, wq_packed, kv_packed, wo_packed, out_bias) in enumerate(cross):
        arr[i].ln_gamma, arr[i].ln_beta, arr[i].ln_eps = _p(gamma), _p(beta), float(eps)
        arr[i].wq_packed, arr[i].kv_packed, arr[i].wo_packed, arr[i].out_bias = _p(wq_packed), _p(kv_packed), _p(wo_packed), _p(out_bias)
    q = _tattn_params(temporal)
    f = _lib.FfParams()
    f.ln_gamma, f.ln_beta, f.ln_eps, f.w_packed, f.up_bias, f.down_bias = _p(ff[0]), _p(ff[1]), float(ff[2]), _p(ff[3]), _p(ff[4]), _p(ff[5])
    pi = None
    if proj_in is not None:
        pi = _lib.ProjInParams()
        pi.gn_scale, pi.gn_shift, pi.w_packed, pi.bias = _p(proj_in[0]), _p(proj_in[1]), _p(proj_in[2]), _p(proj_in[3])
    ev = PROFILER.begin("block_sublayers")
    rc = lib.uav_block_sublayers_f32(_p(x), C.byref(pi) if pi is not None else None, _p(y) if y is not None else None, _p(yh) if yh is not None else None, C.cast(arr, C.c_void_p), lk,
                                     cross_scale, C.byref(q), C.byref(f), n_batch, t_len, hw, XATTN_C, XATTN_HEADS, FF_INNER, temporal_scale,
                                     _stream())
    _lib.check(rc, "uav_block_sublayers_f32")
    PROFILER.end(ev, "block_sublayers" if not PROFILER.detail else f"block_sublayers M={m}",
                 2.0 * m * ((9 if pi is not None else 8) * XATTN_C * XATTN_C) + 2 * 4.0 * m * lk * XATTN_C + 4.0 * m * t_len * XATTN_C
                 + 2.0 * m * XATTN_C * 3 * FF_INNER,
                 4.0 * m * XATTN_C * (1 + (1 if out_f32 else 0) + (1 if out_hilo else 0)))
    if out_f32 and out_hilo:
        return y, yh
    return y if out_f32 else yh


def temporal_attention(qkv, *, n_batch, t_len, hw, c, heads, scale, rope_cos, rope_sin, rot_dim, bias):
    lib = _lib.load()
    _req(qkv, HALF, "qkv")
    out = torch.empty((n_batch * t_len * hw, c), dtype=HALF, device=qkv.device)
    ev = PROFILER.begin("temporal_attention")
    rc = lib.uav_temporal_attention_f16(_p(qkv), _p(out), n_batch, t_len, hw, c, heads, scale, _p(rope_cos),
                                        _p(rope_sin), rot_dim, _p(bias), _stream())
    _lib.check(rc, "uav_temporal_attention_f16")
    PROFILER.end(ev, "temporal_attention", 4.0 * n_batch * hw * t_len * t_len * c, 8.0 * n_batch * t_len * hw * c)
    return out


# ------------------------------------------------------------------------------------------------
def linear_small(x, w, b, *, pre_silu=False, post_silu=False):
    """fp32 x[m<=16][k] @ fp16 w[n][k]^T + b."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(w, HALF, "w")
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    rc = lib.uav_linear_small(_p(x), _p(w), _p(b), _p(y), m, k, n, int(pre_silu), int(post_silu), _stream())
    _lib.check(rc, "uav_linear_small")
    return y


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0):
    lib = _lib.load()
    _req(t, torch.float32, "t")
    out = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    rc = lib.uav_timestep_embedding(_p(t), t.numel(), dim, int(flip_sin_to_cos), freq_shift, _p(out), _stream())
    _lib.check(rc, "uav_timestep_embedding")
    return out


def pack_nhwc(src1, src2=None, c_pad=8, scale=1.0):
    """(B,C,T,H,W) [+ second tensor concatenated on C] -> channels-last fp16 rows [B*T*H*W][c_pad]."""
    lib = _lib.load()
    _req(src1, None, "src1")
    b, c1, t, h, w = src1.shape
    c2 = 0
    if src2 is not None:
        _req(src2, src1.dtype, "src2")
        c2 = src2.shape[1]
    is_f32 = src1.dtype == torch.float32
    if not is_f32 and src1.dtype != HALF:
        raise _lib.UavError("pack_nhwc: fp16 or fp32 input expected")
    dst = torch.empty((b * t * h * w, c_pad), dtype=HALF, device=src1.device)
    rc = lib.uav_pack_nhwc(_p(src1), c1, _p(src2), c2, int(is_f32), _p(dst), c_pad, b, t, h * w, scale, _stream())
    _lib.check(rc, "uav_pack_nhwc")
    return dst


def unpack_ncthw(src, *, c, n_batch, t_len, h, w, out_dtype=HALF, clamp=None):
    lib = _lib.load()
    _req(src, None, "src")
    lo, hi = (-3.0e38, 3.0e38) if clamp is None else clamp
    dst = torch.empty((n_batch, c, t_len, h, w), dtype=out_dtype, device=src.device)
    rc = lib.uav_unpack_ncthw(_p(src), src.shape[-1], int(src.dtype == torch.float32), _p(dst),
                              int(out_dtype == torch.float32), c, n_batch, t_len, h * w, lo, hi, _stream())
    _lib.check(rc, "uav_unpack_ncthw")
    return dst


def cfg_ddim_v0(eps_uncond, eps_text, sample, *, guidance, coef_sample, coef_eps, clip=False, clip_range=1.0):
    """CFG combine + DDIM step_v0; all tensors fp16 (the reference's half arithmetic) or all fp32."""
    lib = _lib.load()
    dt = _sched_dtype(sample, eps_uncond, eps_text)
    g = torch.empty_like(sample); x0 = torch.empty_like(sample)
    fn = lib.uav_cfg_ddim_v0_f32 if dt == torch.float32 else lib.uav_cfg_ddim_v0
    rc = fn(_p(eps_uncond), _p(eps_text), _p(sample), _p(g), _p(x0), sample.numel(), guidance,
            coef_sample, coef_eps, int(clip), clip_range, _stream())
    _lib.check(rc, "uav_cfg_ddim_v0")
    return g, x0


def _sched_dtype(*ts):
    dt = ts[0].dtype
    if dt not in (HALF, torch.float32):
        raise _lib.UavError(f"scheduler tensors must be fp16 or fp32, got {dt}")
    for t in ts:
        if t is not None:
            _req(t, dt, "scheduler tensor")
    return dt


def ddim_vt(x0, guided, sample, *, coef_x0, coef_dir, eps_from_model, eps_from_sample, eps_from_x0=0.0, clip=False,
            clip_range=1.0):
    lib = _lib.load()
    dt = _sched_dtype(sample, x0, guided)
    prev = torch.empty_like(sample)
    fn = lib.uav_ddim_vt_f32 if dt == torch.float32 else lib.uav_ddim_vt
    rc = fn(_p(x0), _p(guided), _p(sample), _p(prev), sample.numel(),
            coef_x0, coef_dir, eps_from_model, eps_from_sample, eps_from_x0, int(clip), clip_range, _stream())
    _lib.check(rc, "uav_ddim_vt")
    return prev


def axpby(x, z, a, b):
    if x.dtype == torch.float32:
        return axpby_f32(x, z, a, b)
    lib = _lib.load()
    y = torch.empty_like(x)
    _lib.check(lib.uav_axpby_f16(_p(_req(x, HALF)), _p(_req(z, HALF)), _p(y), x.numel(), a, b, _stream()), "uav_axpby_f16")
    return y


def cast_f16(x):
    """fp32 rows -> fp16 rows (no-op for fp16 input)."""
    if x.dtype == HALF:
        return x
    lib = _lib.load()
    y = torch.empty(x.shape, dtype=HALF, device=x.device)
    ev = PROFILER.begin("cast_f16")
    _lib.check(lib.uav_cast_f32_f16(_p(_req(x, torch.float32, "x")), _p(y), x.numel(), _stream()), "uav_cast_f32_f16")
    PROFILER.end(ev, "cast_f16", 0.0, 6.0 * x.numel())
    return y


def cast_hilo(x):
    """fp32 rows [M][C] -> fp16 rows [M][2C] = [fp16(x) | fp16(x - fp16(x))]."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    m, c = x.shape
    y = torch.empty((m, 2 * c), dtype=HALF, device=x.device)
    ev = PROFILER.begin("cast_f16")
    _lib.check(lib.uav_cast_f32_hilo(_p(x), _p(y), m, c, _stream()), "uav_cast_f32_hilo")
    PROFILER.end(ev, "cast_f16", 0.0, 8.0 * x.numel())
    return y


def sft_fuse(dec, scale, shift, w, out_f32=False):
    """dec + w*(dec*scale + shift) (Fuse_sft_block); inputs all fp16 or all fp32."""
    lib = _lib.load()
    _req(dec, None, "dec"); _req(scale, dec.dtype, "scale"); _req(shift, dec.dtype, "shift")
    if dec.dtype not in (HALF, torch.float32):
        raise _lib.UavError("sft_fuse: fp16 or fp32 rows expected")
    out = torch.empty(dec.shape, dtype=torch.float32 if out_f32 else HALF, device=dec.device)
    rc = lib.uav_sft_fuse(_p(dec), _p(scale), _p(shift), _p(out), dec.numel(), float(w), int(dec.dtype == torch.float32),
                          int(out_f32), _stream())
    _lib.check(rc, "uav_sft_fuse")
    return out


def propagate_step(feat_prev, feat_cur, flow_prop, flow_check, out, *, c, h, w, feat_chan_stride, flow_chan_stride,
                   nearest, coord_f16, fuse_scale, alpha1, alpha2):
    """One recurrence step of the flow-guided propagation.  The tensors are (possibly strided) views
    of (C,T,H,W) fp16 buffers — or all-fp32 ones (fp32 values, flows and grid arithmetic: the reference's fp32 run) —
    frame planes are addressed in place through the channel strides."""
    lib = _lib.load()
    dt = out.dtype
    for t_ in (feat_prev, feat_cur, flow_prop, flow_check, out):
        if not t_.is_cuda or t_.dtype != dt or dt not in (HALF, torch.float32):
            raise _lib.UavError("propagate_step: fp16 (or all-fp32) GPU tensors expected")
    if dt == torch.float32:
        if coord_f16:
            raise _lib.UavError("propagate_step: fp32 planes are warped on fp32 grids (coord_f16 is the fp16-latent replay)")
        rc = lib.uav_propagate_step_f32(_p(feat_prev), _p(feat_cur), _p(flow_prop), _p(flow_check), _p(out), c, h, w,
                                        feat_chan_stride, flow_chan_stride, int(nearest), fuse_scale, alpha1, alpha2, _stream())
        _lib.check(rc, "uav_propagate_step_f32")
        return out
    rc = lib.uav_propagate_step_f16(_p(feat_prev), _p(feat_cur), _p(flow_prop), _p(flow_check), _p(out), c, h, w,
                                    feat_chan_stride, flow_chan_stride, int(nearest), int(coord_f16), fuse_scale,
                                    alpha1, alpha2, _stream())
    _lib.check(rc, "uav_propagate_step_f16")
    return out


# ------------------------------------------------------------------------------------------------
# fp32 path (RAFT optical flow, K11)
F32 = torch.float32


def pack_conv_f32(weight, bias=None, device=None, cin_pad_to=None):
    """fp32 packing for uav_conv_gemm_f32: [n_pad][k_pad], k = tap*cin_p + c; cin_p = cin rounded up to a
    multiple of 32 (zero weights for the padding channels) or 4 in small mode (cin <= 4)."""
    w = weight.detach().float()
    if w.dim() == 2:
        w = w[:, :, None, None]
    o, i, kh, kw = w.shape
    if cin_pad_to is not None:
        cin_p = cin_pad_to
    elif i <= 4:
        cin_p = 4
    else:
        cin_p = _round_up(i, 32)
    w = w.permute(0, 2, 3, 1)
    if cin_p != i:
        w = torch.nn.functional.pad(w, (0, cin_p - i))
    w = w.reshape(o, kh * kw * cin_p)
    b = None if bias is None else bias.detach().float()
    n = _round_up(o, 4)
    k = w.shape[1]
    n_pad, k_pad = _round_up(n, 128), _round_up(k, 32)
    wp = torch.zeros(n_pad, k_pad, dtype=F32)
    wp[:o, :k] = w
    bp = torch.zeros(n_pad, dtype=F32)
    if b is not None:
        bp[:o] = b
    dev = device if device is not None else weight.device
    return ConvW(wp.to(dev), bp.to(dev), n, n_pad, k_pad, i, cin_p, 1, kh, kw, False)


def conv_gemm_f32(a1, wt: ConvW, *, n_img, hi, wi, stride=1, pad=None, a2=None, residual=None, out_scale=1.0, act=0,
                  out=None):
    """fp32 conv / linear on channels-last rows.  act: 0 none, 1 relu, 2 sigmoid, 4 tanh.  `out` may be a
    column-slice view of a wider row buffer (used to write channel concats in place)."""
    lib = _lib.load()
    _req(a1, F32, "a1")
    c1 = a1.shape[-1]
    c2 = 0 if a2 is None else _req(a2, F32, "a2").shape[-1]
    if c1 + c2 != wt.cin_p:
        raise _lib.UavError(f"conv_f32 input channels {c1}+{c2} != packed {wt.cin_p}")
    if pad is None:
        pad = (wt.kh // 2, wt.kw // 2)
    ph, pw = pad
    ho = (hi + 2 * ph - wt.kh) // stride + 1
    wo = (wi + 2 * pw - wt.kw) // stride + 1
    m = n_img * ho * wo
    if out is None:
        out = torch.empty((m, wt.n), dtype=F32, device=a1.device)
    if out.stride(-1) != 1 or out.dtype != F32 or out.shape[0] != m:
        raise _lib.UavError("conv_f32: bad output view")
    p = _lib.ConvParams()
    p.a1 = _p(a1); p.a2 = _p(a2); p.c1 = c1; p.c2 = c2
    p.w = _p(wt.w); p.bias = _p(wt.bias); p.rowbias = None; p.rows_per_batch = 0; p.rowbias_stride = 0
    p.residual = _p(residual); p.res_stride = 0 if residual is None else residual.stride(0)
    p.out = _p(out); p.out_stride = out.stride(0)
    p.n_img = n_img; p.t_len = 1; p.hi = hi; p.wi = wi; p.ho = ho; p.wo = wo
    p.kt = 1; p.kh = wt.kh; p.kw = wt.kw; p.stride = stride; p.pad_t = 0; p.pad_h = ph; p.pad_w = pw; p.upsample = 0
    p.n = wt.n; p.n_pad = wt.n_pad; p.k_pad = wt.k_pad; p.out_scale = out_scale
    p.flags = {0: 0, 1: _lib.CONV_RELU, 2: _lib.CONV_SIGMOID, 4: _lib.CONV_TANH}[act]
    p.zero_page = _p(zero_page(a1.device))
    ev = PROFILER.begin("conv_gemm_f32")
    _lib.check(lib.uav_conv_gemm_f32(C.byref(p), _stream()), "uav_conv_gemm_f32")
    PROFILER.end(ev, "conv_gemm_f32", 2.0 * m * wt.n * wt.kh * wt.kw * wt.cin, 4.0 * (n_img * hi * wi * wt.cin + m * wt.n))
    return out


def instnorm_f32(x, *, n_img, hw, relu, eps=1e-5):
    lib = _lib.load()
    y = torch.empty_like(_req(x, F32, "x"))
    _lib.check(lib.uav_instnorm_f32(_p(x), _p(y), n_img, hw, x.shape[-1], eps, int(relu), _stream()), "uav_instnorm_f32")
    return y


def add_relu_f32(a, b, relu=True):
    lib = _lib.load()
    o = torch.empty_like(_req(a, F32, "a"))
    _lib.check(lib.uav_add_relu_f32(_p(a), _p(_req(b, F32, "b")), _p(o), a.numel(), int(relu), _stream()), "uav_add_relu_f32")
    return o


def axpby_f32(x, z, a, b, out=None):
    lib = _lib.load()
    y = torch.empty_like(x) if out is None else out
    _lib.check(lib.uav_axpby_f32(_p(_req(x, F32)), _p(_req(z, F32)), _p(y), x.numel(), a, b, _stream()), "uav_axpby_f32")
    return y


def copy_cols_f32(src, src_col, dst, dst_col, ncols, act=0):
    """dst[:, dst_col:dst_col+ncols] = act(src[:, src_col:src_col+ncols]) on row-major fp32 buffers."""
    lib = _lib.load()
    rows = src.shape[0]
    rc = lib.uav_copy_cols_f32(_p(src), src.stride(0), src_col, _p(dst), dst.stride(0), dst_col, ncols, rows, act, _stream())
    _lib.check(rc, "uav_copy_cols_f32")
    return dst


def gru_rh_f32(zr, h):
    lib = _lib.load()
    out = torch.empty_like(h)
    _lib.check(lib.uav_gru_gates_f32(_p(zr), _p(h), None, _p(out), h.shape[0], h.shape[1], 0, _stream()), "uav_gru_gates_f32")
    return out


def gru_blend_f32(zr, q, h):
    lib = _lib.load()
    _lib.check(lib.uav_gru_gates_f32(_p(zr), _p(h), _p(q), _p(h), h.shape[0], h.shape[1], 1, _stream()), "uav_gru_gates_f32")
    return h


def resize_bilinear_f32(x, ho, wo, row0_scale=1.0, row1_scale=1.0):
    """x (..., hi, wi) fp32 contiguous -> (..., ho, wo), bilinear, align_corners=False (F.interpolate semantics)."""
    lib = _lib.load()
    x = _req(x.contiguous(), F32, "x")
    hi, wi = x.shape[-2:]
    planes = x.numel() // (hi * wi)
    out = torch.empty(tuple(x.shape[:-2]) + (ho, wo), dtype=F32, device=x.device)
    _lib.check(lib.uav_resize_bilinear_f32(_p(x), _p(out), planes, hi, wi, ho, wo, float(row0_scale), float(row1_scale),
                                           _stream()), "uav_resize_bilinear_f32")
    return out


def avgpool2_f32(src, src_stride, h, w, p_count):
    lib = _lib.load()
    dst = torch.empty((p_count, (h // 2) * (w // 2)), dtype=F32, device=src.device)
    _lib.check(lib.uav_avgpool2_f32(_p(src), src_stride, h, w, _p(dst), p_count, _stream()), "uav_avgpool2_f32")
    return dst


def corr_lookup_f32(levels, strides, hs, ws, coords, out, radius=4):
    lib = _lib.load()
    ptrs = (C.c_void_p * 4)(*[_p(t) for t in levels])
    st = (C.c_int64 * 4)(*strides)
    hh = (C.c_int32 * 4)(*hs)
    ww = (C.c_int32 * 4)(*ws)
    rc = lib.uav_corr_lookup_f32(ptrs, st, hh, ww, _p(coords), coords.stride(0), _p(out), out.stride(0), coords.shape[0],
                                 radius, _stream())
    _lib.check(rc, "uav_corr_lookup_f32")
    return out


def convex_upsample_f32(flow_rows, mask_rows, n, h, w):
    lib = _lib.load()
    out = torch.empty((n, 2, 8 * h, 8 * w), dtype=F32, device=flow_rows.device)
    rc = lib.uav_convex_upsample_f32(_p(flow_rows), flow_rows.stride(0), _p(mask_rows), _p(out), n, h, w, _stream())
    _lib.check(rc, "uav_convex_upsample_f32")
    return out


# ------------------------------------------------------------------------------------------------
# colour correction (K12): fp32 planes (T*C, H, W)
def _planes(x):
    x = _req(x.contiguous(), F32, "x")
    if x.dim() < 2:
        raise _lib.UavError("expected (..., H, W)")
    h, w = x.shape[-2:]
    return x, x.numel() // (h * w), h, w


def plane_stats_f32(x):
    """Per-plane mean and unbiased variance of (..., H, W) fp32 -> two fp32 vectors [planes]."""
    lib = _lib.load()
    x, planes, h, w = _planes(x)
    mean = torch.empty(planes, dtype=F32, device=x.device); var = torch.empty_like(mean)
    nb = lib.uav_plane_stats_workspace_bytes(planes)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    _lib.check(lib.uav_plane_stats_f32(_p(x), planes, h * w, _p(mean), _p(var), _p(ws), nb, _stream()), "uav_plane_stats_f32")
    return mean, var


def adain_apply_f32(x, c_mean, c_var, s_mean, s_var, eps=1e-5):
    lib = _lib.load()
    x, planes, h, w = _planes(x)
    out = torch.empty_like(x)
    rc = lib.uav_adain_apply_f32(_p(x), _p(out), planes, h * w, _p(c_mean), _p(c_var), _p(s_mean), _p(s_var), float(eps), _stream())
    _lib.check(rc, "uav_adain_apply_f32")
    return out


def atrous_blur_f32(x, radius, high=None):
    """3x3 a-trous blur (dilation `radius`, replicate padding); `high` (optional, same shape) accumulates x - blur(x) in place."""
    lib = _lib.load()
    x, planes, h, w = _planes(x)
    low = torch.empty_like(x)
    if high is not None:
        _req(high, F32, "high")
    _lib.check(lib.uav_atrous_blur_f32(_p(x), _p(low), _p(high), planes, h, w, int(radius), _stream()), "uav_atrous_blur_f32")
    return low


def resize_bicubic_f32(x, ho, wo, scale_h=None, scale_w=None):
    """F.interpolate(x, mode='bicubic', align_corners=False): pass scale_* = 1/scale_factor when a scale factor was given
    (ATen uses it verbatim), else in/out."""
    lib = _lib.load()
    x, planes, hi, wi = _planes(x)
    out = torch.empty(tuple(x.shape[:-2]) + (ho, wo), dtype=F32, device=x.device)
    sh = hi / ho if scale_h is None else scale_h
    sw = wi / wo if scale_w is None else scale_w
    _lib.check(lib.uav_resize_bicubic_f32(_p(x), _p(out), planes, hi, wi, ho, wo, float(sh), float(sw), _stream()), "uav_resize_bicubic_f32")
    return out


def resize_area_f32(x, ho, wo, mul=1.0):
    """F.interpolate(x, (ho, wo), mode='area') * mul on (..., H, W) fp32."""
    lib = _lib.load()
    x, planes, hi, wi = _planes(x)
    out = torch.empty(tuple(x.shape[:-2]) + (ho, wo), dtype=F32, device=x.device)
    _lib.check(lib.uav_resize_area_f32(_p(x), _p(out), planes, hi, wi, ho, wo, float(mul), _stream()), "uav_resize_area_f32")
    return out


# ------------------------------------------------------------------------------------------------
# frame I/O conversions of the CLI (K13)
def frames_to_clip_f32(frames):
    """(T,C,H,W) uint8 / fp32 frames in 0..255 -> (C,T,H,W) fp32 in [-1,1] (`(x/255. - 0.5)*2` + the CLI's rearrange)."""
    lib = _lib.load()
    if frames.dtype not in (torch.uint8, torch.float32) or frames.dim() != 4:
        raise _lib.UavError("frames_to_clip_f32: (T,C,H,W) uint8 or fp32 expected")
    _req(frames, frames.dtype, "frames")
    t, c, h, w = frames.shape
    out = torch.empty((c, t, h, w), dtype=torch.float32, device=frames.device)
    rc = lib.uav_frames_to_clip_f32(_p(frames), int(frames.dtype == torch.uint8), _p(out), t, c, h * w, _stream())
    _lib.check(rc, "uav_frames_to_clip_f32")
    return out


def clip_to_frames_u8(frames):
    """(T,C,H,W) fp32 in [-1,1] -> (T,H,W,C) uint8 (`(x/2 + 0.5).clamp(0,1)*255`, truncated like numpy's astype)."""
    lib = _lib.load()
    _req(frames, torch.float32, "frames")
    t, c, h, w = frames.shape
    out = torch.empty((t, h, w, c), dtype=torch.uint8, device=frames.device)
    _lib.check(lib.uav_clip_to_frames_u8(_p(frames), _p(out), t, c, h * w, _stream()), "uav_clip_to_frames_u8")
    return out
