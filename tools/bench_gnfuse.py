"""A/B of the GroupNorm statistics fused into the conv epilogue (GPU box only): per shape, the conv alone, the conv
writing partials, and the GroupNorm statistics pass it replaces (stand-alone partial + finalize vs finalize of the conv's
partials).  One text line per case."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from uav import ops  # noqa: E402
from bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
CASES = [  # name, n_img, t_len, h, w, cin, cout, k3, residual, temb, f32
    ("3x3 512->512 @16x160x160 +res", 16, 8, 160, 160, 512, 512, (1, 3, 3), True, False, False),
    ("3x3 512->512 @16x160x160 +temb", 16, 8, 160, 160, 512, 512, (1, 3, 3), False, True, False),
    ("3x3 256->256 @16x320x320 +res", 16, 8, 320, 320, 256, 256, (1, 3, 3), True, False, False),
    ("1x1 512->512 M=409600 +res", 16, 8, 160, 160, 512, 512, (1, 1, 1), True, False, False),
    ("t3 512->512 @16x160x160 +res", 16, 8, 160, 160, 512, 512, (3, 1, 1), True, False, False),
    ("3x3 1024->1024 @16x40x40 +res", 16, 8, 40, 40, 1024, 1024, (1, 3, 3), True, False, False),
    ("vae 3x3 128->128 @3x1280x1280 f32 +res", 3, 3, 1280, 1280, 128, 128, (1, 3, 3), True, False, True),
    ("vae 3x3 256->256 @3x640x640 f32 +res", 3, 3, 640, 640, 256, 256, (1, 3, 3), True, False, True),
]
for name, n_img, t_len, h, w, cin, cout, k3, res, temb, f32 in CASES:
    m = n_img * h * w
    x = torch.randn(m, cin, device=dev).half()
    wt = torch.randn(cout, cin, *k3) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5
    cw = ops.pack_conv(wt, torch.zeros(cout), device=dev)
    rr = (torch.randn(m, cout, device=dev) if f32 else torch.randn(m, cout, device=dev).half()) if res else None
    rb = torch.randn(n_img // t_len, cout, device=dev) if temb else None
    kw = dict(n_img=n_img, t_len=t_len, hi=h, wi=w, residual=rr, rowbias=rb, rows_per_batch=t_len * h * w, out_f32=f32)
    gamma = torch.ones(cout, device=dev); beta = torch.zeros(cout, device=dev)
    gkw = dict(n_inst=n_img // t_len, rows_per_inst=t_len * h * w, groups=32, eps=1e-6)
    t0 = timeit(lambda: ops.conv_gemm(x, cw, **kw), 5)
    t1 = timeit(lambda: ops.conv_gemm(x, cw, gn_groups=32, **kw), 5)
    y0 = ops.conv_gemm(x, cw, **kw)
    y1 = ops.conv_gemm(x, cw, gn_groups=32, **kw)
    fused = getattr(y1, "_uav_gn", None) is not None
    s0 = timeit(lambda: ops.groupnorm_scale_shift(y0, gamma, beta, **gkw), 5)
    s1 = timeit(lambda: ops.groupnorm_scale_shift(y1, gamma, beta, **gkw), 5)
    print(f"{name:42s} conv {t0 * 1e3:7.3f} -> {t1 * 1e3:7.3f} ms ({(t1 / t0 - 1) * 100:+5.1f} %)   stats {s0 * 1e3:6.3f} -> "
          f"{s1 * 1e3:6.3f} ms   net {((t1 + s1) - (t0 + s0)) * 1e3:+7.3f} ms   fused={fused}", flush=True)
    del x, y0, y1, rr
