"""Same-process, interleaved A/B of the four-wave conv kernel (conv_gemm256w_kernel, default) against the 8-wave kernel
(UAV_CONV_NO_W4) on the conv / linear shapes of BASELINE configs[1] with the epilogues the UNet uses on them.
One JSON line per case: median ms and TFLOP/s of both.  usage: python tools/bench_w4.py [case-substring ...]"""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")
ONLY = sys.argv[1:]
ROUNDS, PER = int(os.environ.get("UAV_AB_ROUNDS", "5")), int(os.environ.get("UAV_AB_PER", "4"))


def silu_like(*shape):
    return torch.nn.functional.silu(torch.randn(*shape, device=dev)).half()


def time_once(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, n_img, t_len, h, w, cin, cout, k3, *, res=None, out_f32=False, gn=None, geglu=False, c2=0, rowbias=False, stride=1,
         pad=None, out_hw=None):
    if ONLY and not any(o in name for o in ONLY):
        return
    if w == 1 and h >= 65536:                    # token rows of a linear: the kernel packs (y, x) in 16 bits each
        n_img, h = ops._factor_rows(h)
    rows = n_img * h * w
    x = silu_like(rows, cin - c2)
    x2 = silu_like(rows, c2) if c2 else None
    wt = torch.randn(cout, cin, *k3) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5
    cw = ops.pack_conv(wt, 0.1 * torch.randn(cout), geglu=geglu, device=dev)
    n_out = cout // 2 if geglu else cout
    if out_hw:
        ho, wo = out_hw
    else:
        pd = pad or (k3[0] // 2, k3[1] // 2, k3[2] // 2)
        ho, wo = (h + 2 * pd[1] - k3[1]) // stride + 1, (w + 2 * pd[2] - k3[2]) // stride + 1
    m = n_img * ho * wo
    r = None if res is None else (torch.randn(m, n_out, device=dev) if res == "f32" else torch.randn(m, n_out, device=dev).half())
    rb = torch.randn(n_img // t_len, cout, device=dev) if rowbias else None
    out = torch.empty((m, n_out), dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    kw = dict(a2=x2, n_img=n_img, t_len=t_len, hi=h, wi=w, stride=stride, pad=pad, out_hw=out_hw, residual=r, out_f32=out_f32,
              gn_groups=gn, out=out, rowbias=rb, rows_per_batch=t_len * ho * wo if rb is not None else 0)
    fns = {"wave8": lambda: ops.conv_gemm(x, cw, no_w4=True, **kw), "wave4": lambda: ops.conv_gemm(x, cw, **kw)}
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    t = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            t[k].append(time_once(f, PER))
    fl = 2.0 * m * cout * cin * k3[0] * k3[1] * k3[2]
    d = {"case": name}
    for k in fns:
        med = statistics.median(t[k])
        d[k] = {"ms": round(med, 4), "tflops": round(fl / med / 1e9, 1)}
    d["speedup"] = round(d["wave8"]["ms"] / d["wave4"]["ms"], 3)
    print(json.dumps(d), flush=True)


def main():
    C = dict(n_img=16, t_len=8)
    L = dict(n_img=1, t_len=1, w=1)
    case("3x3 512->512 @160 rowbias->f32+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), rowbias=True, gn=32, out_f32=True)
    case("3x3 512->512 @160 res32->f32+gn", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), res="f32", out_f32=True, gn=32)
    case("3x3 256->256 @320 rowbias->f32+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 3, 3), rowbias=True, gn=32, out_f32=True)
    case("3x3 256->256 @320 res32->f32+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 3, 3), res="f32", out_f32=True, gn=32)
    case("3x3 1024->1024 @40 rowbias->f32+gn", **C, h=40, w=40, cin=1024, cout=1024, k3=(1, 3, 3), rowbias=True, gn=32, out_f32=True)
    case("3x3 1024->512 @160 cat", **C, h=160, w=160, cin=1024, cout=512, k3=(1, 3, 3), c2=512, rowbias=True, gn=32, out_f32=True)
    case("3x3 768->256 @320 cat", **C, h=320, w=320, cin=768, cout=256, k3=(1, 3, 3), c2=256, rowbias=True, gn=32, out_f32=True)
    case("3x3 512->512 @80 rowbias->f32+gn", **C, h=80, w=80, cin=512, cout=512, k3=(1, 3, 3), rowbias=True, gn=32, out_f32=True)
    case("3x3 512->512 s2 @160->80", **C, h=160, w=160, cin=512, cout=512, k3=(1, 3, 3), stride=2, out_f32=True)
    case("2x2 512->512 @160 phase", **C, h=160, w=160, cin=512, cout=512, k3=(1, 2, 2), pad=(0, 1, 1), out_hw=(160, 160), out_f32=True)
    case("t3 512->512 @160 res16+gn", **C, h=160, w=160, cin=512, cout=512, k3=(3, 1, 1), res="f16", gn=32)
    case("t3 512->512 @160 res32->f32", **C, h=160, w=160, cin=512, cout=512, k3=(3, 1, 1), res="f32", out_f32=True)
    case("t3 1024->1024 @40 res32->f32", **C, h=40, w=40, cin=1024, cout=1024, k3=(3, 1, 1), res="f32", out_f32=True)
    case("t5 512->512 @320", **C, h=320, w=320, cin=512, cout=512, k3=(5, 1, 1), res="f32", out_f32=True)
    case("t3 256->256 @320 res32->f32", **C, h=320, w=320, cin=256, cout=256, k3=(3, 1, 1), res="f32", out_f32=True)
    case("lin 512->512 M=409600 ->f16", **L, h=409600, cin=512, cout=512, k3=(1, 1, 1))
    case("lin 512->512 M=409600 res32->f32", **L, h=409600, cin=512, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("lin 512->512 M=409600 ->f32", **L, h=409600, cin=512, cout=512, k3=(1, 1, 1), out_f32=True)
    case("lin 512->1536 M=409600", **L, h=409600, cin=512, cout=1536, k3=(1, 1, 1))
    case("lin 512->4096 geglu M=409600", **L, h=409600, cin=512, cout=4096, k3=(1, 1, 1), geglu=True)
    case("lin 2048->512 M=409600 res32->f32", **L, h=409600, cin=2048, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("lin 1024->512 M=409600 hilo res32->f32", **L, h=409600, cin=1024, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("lin 512->512 M=102400 res32->f32", **L, h=102400, cin=512, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("lin 512->1536 M=102400", **L, h=102400, cin=512, cout=1536, k3=(1, 1, 1))
    case("lin 512->4096 geglu M=102400", **L, h=102400, cin=512, cout=4096, k3=(1, 1, 1), geglu=True)
    case("lin 2048->512 M=102400 res32->f32", **L, h=102400, cin=2048, cout=512, k3=(1, 1, 1), res="f32", out_f32=True)
    case("lin 1024->1024 M=25600 res32->f32", **L, h=25600, cin=1024, cout=1024, k3=(1, 1, 1), res="f32", out_f32=True)
    case("lin 1024->3072 M=25600", **L, h=25600, cin=1024, cout=3072, k3=(1, 1, 1))
    case("lin 1024->8192 geglu M=25600", **L, h=25600, cin=1024, cout=8192, k3=(1, 1, 1), geglu=True)
    case("lin 4096->1024 M=25600 res32->f32", **L, h=25600, cin=4096, cout=1024, k3=(1, 1, 1), res="f32", out_f32=True)
    case("1x1 256->256 @320 res32->f32+gn", **C, h=320, w=320, cin=256, cout=256, k3=(1, 1, 1), res="f32", out_f32=True, gn=32)
    case("1x1 768->256 @320 cat ->f32+gn", **C, h=320, w=320, cin=768, cout=256, k3=(1, 1, 1), c2=256, out_f32=True, gn=32)
    case("3x3 128->128 @3x1280 vae", n_img=3, t_len=3, h=1280, w=1280, cin=128, cout=128, k3=(1, 3, 3), out_f32=True, gn=32)
    case("3x3 256->256 @3x640 vae", n_img=3, t_len=3, h=640, w=640, cin=256, cout=256, k3=(1, 3, 3), out_f32=True, gn=32)


if __name__ == "__main__":
    main()
