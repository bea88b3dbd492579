"""TEST INFRASTRUCTURE ONLY — lets the fp32 oracle (oracle/uav_oracle.py) run ON the GPU as an independent checker at
sizes the CPU cannot afford (BASELINE configs[1]: UNet forward on (2,4,8,320,320), one VAE decode chunk at 320x320).

The oracle is plain torch.  On the GPU its convolutions would go to MIOpen, which has no pre-compiled kernels for gfx950
in this image (every new shape is a run-time compile, minutes in total); `F.conv2d` / `F.conv3d` are therefore swapped —
inside the oracle module only — for a tap-by-tap formulation on fp32 GEMMs (rocBLAS / hipBLASLt, exact-fp32 MFMA):

    y[n, :, o] = bias + sum_taps  x_padded[n, o*stride + tap, :] @ W[:, :, tap]^T

i.e. the same arithmetic as a direct convolution, fp32 products and fp32 accumulation, only the summation order differs
(~1e-6 relative).  Everything else (GroupNorm, softmax, interpolate, ...) is ATen's native fp32 kernels.  Factory calls
inside the oracle (`torch.arange`, ...) follow the `torch.device(dev)` context.  Never imported by the product path.
"""
import contextlib
import itertools

import torch
import torch.nn.functional as F


def conv_nd_taps(x, w, b=None, stride=1, padding=0):
    """N-d convolution (2-d or 3-d, cross-correlation like F.convNd, dilation 1, groups 1) as one GEMM per tap."""
    nd = x.dim() - 2
    stride = (stride,) * nd if isinstance(stride, int) else tuple(stride)
    padding = (padding,) * nd if isinstance(padding, int) else tuple(padding)
    assert w.dim() == nd + 2 and x.shape[1] == w.shape[1], (x.shape, w.shape)
    n, c = x.shape[:2]
    o = w.shape[0]
    k = tuple(w.shape[2:])
    sp_in = tuple(x.shape[2:])
    sp_out = tuple((sp_in[d] + 2 * padding[d] - k[d]) // stride[d] + 1 for d in range(nd))
    pad = []
    for d in reversed(range(nd)):
        pad += [padding[d], padding[d]]
    xp = F.pad(x, pad) if any(padding) else x
    perm = (0,) + tuple(range(2, nd + 2)) + (1,)
    xp = xp.permute(perm).contiguous()                               # channels-last (N, *sp, C)
    rows = n
    for s_ in sp_out:
        rows *= s_
    out = torch.zeros((rows, o), dtype=torch.float32, device=x.device) if b is None else \
        b.float().reshape(1, o).expand(rows, o).contiguous()
    for tap in itertools.product(*[range(kk) for kk in k]):
        sl = (slice(None),) + tuple(slice(tap[d], tap[d] + (sp_out[d] - 1) * stride[d] + 1, stride[d]) for d in range(nd))
        xs = xp[sl].reshape(rows, c)
        wt = w[(slice(None), slice(None)) + tap].float().t().contiguous()       # (C, O)
        out.addmm_(xs, wt)
    inv = (0, nd + 1) + tuple(range(1, nd + 1))
    return out.reshape((n,) + sp_out + (o,)).permute(inv)


class _F:
    """torch.nn.functional with conv2d / conv3d replaced; every other attribute is the real one."""

    def __getattr__(self, name):
        return getattr(F, name)

    @staticmethod
    def conv2d(x, w, b=None, stride=1, padding=0):
        return conv_nd_taps(x, w, b, stride, padding)

    @staticmethod
    def conv3d(x, w, b=None, stride=1, padding=0):
        return conv_nd_taps(x, w, b, stride, padding)


@contextlib.contextmanager
def oracle_on(dev):
    """Run oracle functions on `dev`: GEMM-based convolutions, factory calls default to `dev`, no TF32."""
    import uav_oracle as O
    saved = O.F
    tf32 = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    O.F = _F()
    try:
        with torch.device(dev), torch.no_grad():
            yield O
    finally:
        O.F = saved
        torch.backends.cuda.matmul.allow_tf32 = tf32
