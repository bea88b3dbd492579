"""Conv / linear micro-benchmarks WITH the epilogue variants of the hot path (bias, time-embedding row, fp16 / fp32 residual,
fp32 output, GroupNorm statistics, GEGLU) at the shapes of BASELINE configs[1].  One JSON line per case; used for same-box
A/Bs of two library builds (UAV_HIP_LIB)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=6, warm=2):
    iters *= int(os.environ.get("UAV_EPI_ITERS_X", "1"))
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


ONLY = sys.argv[1:]          # optional: run only the cases whose name contains one of these substrings


def case(name, n_img, t_len, h, w, cin, cout, k3, *, res=None, out_f32=False, rowbias=False, gn=None, geglu=False, iters=6):
    if ONLY and not any(o in name for o in ONLY):
        return
    m = n_img * h * w
    x = torch.randn(m, cin, device=dev).half()
    wt = torch.randn(cout, cin, *k3) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5
    cw = ops.pack_conv(wt, 0.1 * torch.randn(cout), geglu=geglu, device=dev)
    n_out = cout // 2 if geglu else cout
    r = None
    if res == "f16":
        r = torch.randn(m, n_out, device=dev).half()
    elif res == "f32":
        r = torch.randn(m, n_out, device=dev)
    rb = torch.randn(n_img // t_len, cout, device=dev) if rowbias else None
    out = torch.empty((m, n_out), dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    fn = lambda: ops.conv_gemm(x, cw, n_img=n_img, t_len=t_len, hi=h, wi=w, residual=r, out_f32=out_f32, rowbias=rb,
                               rows_per_batch=t_len * h * w, gn_groups=gn, out=out)
    s = timeit(fn, iters)
    fl = 2.0 * m * cout * cin * k3[0] * k3[1] * k3[2]
    print(json.dumps({"case": name, "ms": s * 1e3, "tflops": fl / s / 1e12}), flush=True)


def main():
    L = dict(n_img=640, t_len=1, h=640, w=1)                  # token rows M = 409600
    case("linear 512->512 M=409600 bias", **L, cin=512, cout=512, k3=(1, 1, 1))
    case("linear 512->512 M=409600 bias+res16", **L, cin=512, cout=512, k3=(1, 1, 1), res="f16")
    case("linear 512->512 M=409600 bias+res16+gn", **L, cin=512, cout=512, k3=(1, 1, 1), res="f16", gn=32)
    case("linear 512->512 M=409600 bias+res32->f32", **L, cin=512, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("linear 512->512 M=409600 bias+res32->f16", **L, cin=512, cout=512, k3=(1, 1, 1), res="f32")
    case("linear 512->512 M=409600 bias+res32->f32+gn", **L, cin=512, cout=512, k3=(1, 1, 1), res="f32", out_f32=True, gn=32)
    case("linear 512->1536 M=409600 nobias-like", **L, cin=512, cout=1536, k3=(1, 1, 1))
    case("linear 512->4096 geglu M=409600", **L, cin=512, cout=4096, k3=(1, 1, 1), geglu=True, iters=4)
    case("linear 2048->512 M=409600 bias+res16", **L, cin=2048, cout=512, k3=(1, 1, 1), res="f16")
    case("linear 2048->512 M=409600 bias+res32->f32", **L, cin=2048, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("linear 1024->1024 M=25600 bias+res16", n_img=40, t_len=1, h=640, w=1, cin=1024, cout=1024, k3=(1, 1, 1), res="f16")
    C = dict(n_img=16, t_len=8)
    case("3x3 512->512 @16x160x160 bias+rowbias+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), rowbias=True, gn=32)
    case("3x3 512->512 @16x160x160 bias+rowbias->f32+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), rowbias=True, gn=32, out_f32=True)
    case("3x3 512->512 @16x160x160 bias+res16+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), res="f16", gn=32)
    case("3x3 512->512 @16x160x160 bias+res32->f32+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), res="f32", out_f32=True, gn=32)
    case("3x3 256->256 @16x320x320 bias+rowbias+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 3, 3), rowbias=True, gn=32, iters=4)
    case("3x3 256->256 @16x320x320 bias+res16+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 3, 3), res="f16", gn=32, iters=4)
    case("3x3 256->256 @16x320x320 bias+res32->f32+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 3, 3), res="f32", out_f32=True, gn=32, iters=4)
    case("t3 512->512 @16x160x160 bias+res16+gn", **C, h=160, w=160, cin=512, cout=512, k3=(3, 1, 1), res="f16", gn=32)
    case("1x1 512->512 @16x160x160 bias+res32->f32", **C, h=160, w=160, cin=512, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("1x1 512->512 @16x160x160 bias+res32->f32+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 1, 1), res="f32", out_f32=True, gn=32)
    case("1x1 512->512 @16x160x160 bias+res16+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 1, 1), res="f16", gn=32)
    case("1x1 256->256 @16x320x320 bias+res32->f32+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 1, 1), res="f32", out_f32=True, gn=32, iters=4)


if __name__ == "__main__":
    main()
