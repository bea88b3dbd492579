"""Micro-benchmarks of the individual HIP kernels at the north-star shapes (GPU box only).
Prints one JSON line per case: achieved TFLOP/s (MFMA-bound kernels) or GB/s (HBM-bound)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def conv_case(name, n_img, t_len, h, w, cin, cout, k3, stride=1, ups=False, iters=5):
    x = torch.randn(n_img * h * w, cin, device=dev).half()
    wt = torch.randn(cout, cin, *k3) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5
    cw = ops.pack_conv(wt, torch.zeros(cout), device=dev)
    fn = lambda: ops.conv_gemm(x, cw, n_img=n_img, t_len=t_len, hi=h, wi=w, stride=stride, upsample=ups)
    s = timeit(fn, iters)
    ho, wo = (2 * h, 2 * w) if ups else (h // stride, w // stride)
    fl = 2.0 * n_img * ho * wo * cout * cin * k3[0] * k3[1] * k3[2]
    print(json.dumps({"kernel": "conv_gemm", "case": name, "ms": s * 1e3, "tflops": fl / s / 1e12}), flush=True)


def main():
    which = sys.argv[1:] or ["conv", "norm", "attn", "tattn"]
    if "conv" in which:
        conv_case("3x3 256->256 @16x320x320", 16, 8, 320, 320, 256, 256, (1, 3, 3))
        conv_case("3x3 512->512 @16x160x160", 16, 8, 160, 160, 512, 512, (1, 3, 3))
        conv_case("3x3 1024->1024 @16x40x40", 16, 8, 40, 40, 1024, 1024, (1, 3, 3))
        conv_case("3x3 512->512 @16x320x320", 16, 8, 320, 320, 512, 512, (1, 3, 3), iters=3)
        conv_case("t3 512->512 @16x160x160", 16, 8, 160, 160, 512, 512, (3, 1, 1))
        conv_case("linear 512->512 M=409600", 640, 1, 640, 1, 512, 512, (1, 1, 1))
        conv_case("linear 512->4096 M=409600", 640, 1, 640, 1, 512, 4096, (1, 1, 1), iters=3)
        conv_case("linear 2048->512 M=409600", 640, 1, 640, 1, 2048, 512, (1, 1, 1))
        conv_case("3x3 128->128 @3x1280x1280", 3, 3, 1280, 1280, 128, 128, (1, 3, 3), iters=3)
        conv_case("3x3 1024->512 @16x160x160", 16, 8, 160, 160, 1024, 512, (1, 3, 3), iters=3)
        conv_case("t5 512->512 @16x320x320", 16, 8, 320, 320, 512, 512, (5, 1, 1), iters=3)
    if "norm" in which:
        for c, rows in [(256, 16 * 320 * 320), (512, 16 * 160 * 160), (1024, 16 * 40 * 40)]:
            x = torch.randn(rows, c, device=dev).half()
            g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
            s = timeit(lambda: ops.groupnorm_scale_shift(x, g, b, n_inst=2, rows_per_inst=rows // 2, groups=32, eps=1e-5))
            print(json.dumps({"kernel": "gn_stats", "c": c, "rows": rows, "ms": s * 1e3, "GBps": rows * c * 2 / s / 1e9}), flush=True)
            sc, sh = ops.groupnorm_scale_shift(x, g, b, n_inst=2, rows_per_inst=rows // 2, groups=32, eps=1e-5)
            s = timeit(lambda: ops.groupnorm_apply(x, sc, sh, n_inst=2, rows_per_inst=rows // 2, silu=True))
            print(json.dumps({"kernel": "gn_apply", "c": c, "rows": rows, "ms": s * 1e3, "GBps": rows * c * 4 / s / 1e9}), flush=True)
            s = timeit(lambda: ops.layernorm(x, g, b))
            print(json.dumps({"kernel": "layernorm", "c": c, "rows": rows, "ms": s * 1e3, "GBps": rows * c * 4 / s / 1e9}), flush=True)
    if "attn" in which:
        for d, heads, bq, lq, lk, qpk in [(128, 8, 16, 1600, 1600, 1), (64, 8, 16, 25600, 77, 8), (128, 8, 16, 1600, 77, 8),
                                          (512, 1, 1, 25600, 25600, 1)]:
            c = heads * d
            q = torch.randn(bq * lq, c, device=dev).half()
            k = torch.randn(bq // qpk * lk, c, device=dev).half(); v = torch.randn_like(k)
            s = timeit(lambda: ops.attention(q, k, v, bq=bq, lq=lq, lk=lk, heads=heads, head_dim=d, q_per_kv=qpk), iters=3)
            fl = 4.0 * bq * heads * lq * lk * d
            print(json.dumps({"kernel": "attention", "d": d, "lq": lq, "lk": lk, "ms": s * 1e3, "tflops": fl / s / 1e12,
                              "GBps": (2 * bq * lq * c * 2) / s / 1e9}), flush=True)
    if "tattn" in which:
        for c, hw in [(512, 160 * 160), (1024, 40 * 40)]:
            heads, t_len, nb = 8, 8, 2
            qkv = torch.randn(nb * t_len * hw, 3 * c, device=dev).half()
            bias = torch.zeros(heads, t_len, t_len, device=dev)
            cos = torch.ones(t_len, 16, device=dev); sin = torch.zeros(t_len, 16, device=dev)
            s = timeit(lambda: ops.temporal_attention(qkv, n_batch=nb, t_len=t_len, hw=hw, c=c, heads=heads, scale=0.125,
                                                      rope_cos=cos, rope_sin=sin, rot_dim=32, bias=bias))
            print(json.dumps({"kernel": "temporal_attention", "c": c, "hw": hw, "ms": s * 1e3,
                              "GBps": nb * t_len * hw * c * 8 / s / 1e9}), flush=True)


if __name__ == "__main__":
    main()
