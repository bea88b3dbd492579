"""Same-process, interleaved A/B of the short-K kernel (conv_gemm_sk_kernel) against the general 256x256 kernel
(UAV_CONV_NO_SHORTK) on the 1x1 / linear shapes of BASELINE configs[1] with the epilogues the UNet uses on them.
One JSON line per case: median / min ms and TFLOP/s of both, and the HBM rate of the algorithmic bytes.
usage: python tools/bench_shortk.py [case-substring ...]        (UAV_CONV_SK=2 selects the compiler-scheduled k-step)"""
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
ROUNDS, PER = int(os.environ.get("UAV_SK_ROUNDS", "5")), int(os.environ.get("UAV_SK_PER", "6"))


def silu_like(*shape):
    return torch.nn.functional.silu(torch.randn(*shape, device=dev)).half()


def time_once(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, m, cin, cout, *, res=None, out_f32=False, gn=None, geglu=False, c2=0):
    if ONLY and not any(o in name for o in ONLY):
        return
    x = silu_like(m, cin - c2)
    x2 = silu_like(m, c2) if c2 else None
    wt = torch.randn(cout, cin, 1, 1, 1) * cin ** -0.5
    cw = ops.pack_conv(wt, 0.1 * torch.randn(cout), geglu=geglu, device=dev)
    n_out = cout // 2 if geglu else cout
    r = None if res is None else (torch.randn(m, n_out, device=dev) if res == "f32" else torch.randn(m, n_out, device=dev).half())
    out = torch.empty((m, n_out), dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    n_img, hi = ops._factor_rows(m)
    kw = dict(a2=x2, n_img=n_img, t_len=1, hi=hi, wi=1, residual=r, out_f32=out_f32, gn_groups=gn, out=out)
    fns = {"general": lambda: ops.conv_gemm(x, cw, no_shortk=True, **kw), "shortk": lambda: ops.conv_gemm(x, cw, **kw)}
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    t = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            t[k].append(time_once(f, PER))
    fl = 2.0 * m * cout * cin
    nbytes = 2.0 * m * cin + 2.0 * cout * cin + (4.0 if out_f32 else 2.0) * m * n_out + (0 if r is None else r.element_size() * m * n_out)
    d = {"case": name}
    for k in fns:
        med, mn = statistics.median(t[k]), min(t[k])
        d[k] = {"ms_median": round(med, 4), "ms_min": round(mn, 4), "tflops": round(fl / med / 1e9, 1), "GBps": round(nbytes / med / 1e6, 0)}
    d["speedup"] = round(d["general"]["ms_median"] / d["shortk"]["ms_median"], 3)
    print(json.dumps(d), flush=True)


def main():
    M = 409600
    case("512->512 M=409600 bias->f16 (to_q)", M, 512, 512)
    case("512->512 M=409600 res32->f32 (to_out / proj_out)", M, 512, 512, res="f32", out_f32=True)
    case("512->512 M=409600 res32->f32+gn", M, 512, 512, res="f32", out_f32=True, gn=32)
    case("512->512 M=409600 ->f32 (proj_in)", M, 512, 512, out_f32=True)
    case("512->512 M=409600 res32->f16 (block tail)", M, 512, 512, res="f32")
    case("512->512 M=409600 res16->f16+gn", M, 512, 512, res="f16", gn=32)
    case("512->1536 M=409600 qkv", M, 512, 1536)
    case("512->4096 geglu M=409600", M, 512, 4096, geglu=True)
    case("1024->512 M=409600 res32->f32", M, 1024, 512, res="f32", out_f32=True)
    case("768->512 M=409600 cat ->f32+gn", M, 768, 512, out_f32=True, gn=32, c2=256)
    case("512->512 M=102400 res32->f32", 102400, 512, 512, res="f32", out_f32=True)
    case("512->512 M=102400 bias->f16", 102400, 512, 512)
    case("512->1536 M=102400 qkv", 102400, 512, 1536)
    case("512->4096 geglu M=102400", 102400, 512, 4096, geglu=True)
    case("1024->1024 M=25600 res32->f32", 25600, 1024, 1024, res="f32", out_f32=True)
    case("1024->1024 M=25600 bias->f16", 25600, 1024, 1024)
    case("1024->3072 M=25600 qkv", 25600, 1024, 3072)
    case("1024->8192 geglu M=25600", 25600, 1024, 8192, geglu=True)
    case("256->256 M=1638400 res32->f32+gn", 1638400, 256, 256, res="f32", out_f32=True, gn=32)
    case("512->256 M=1638400 ->f32+gn", 1638400, 512, 256, out_f32=True, gn=32)
    case("768->256 M=1638400 cat ->f32+gn", 1638400, 768, 256, out_f32=True, gn=32, c2=256)
    case("512->512 M=1638400 res32->f32", 1638400, 512, 512, res="f32", out_f32=True)
    case("256->512 M=409600 ->f32", M, 256, 512, out_f32=True)


if __name__ == "__main__":
    main()
