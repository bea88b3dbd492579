"""Development: phase time stamps (s_memtime ticks) of conv_gemm256w_kernel per workgroup on a few shapes.
usage: UAV_CONV_W4_TRACE=1 python tools/trace_w4.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(name, n_img, t_len, h, w, cin, cout, k3, res=None, out_f32=False, geglu=False):
    if w == 1 and h >= 65536:
        n_img, h = ops._factor_rows(h)
    rows = n_img * h * w
    x = torch.nn.functional.silu(torch.randn(rows, cin, device=dev)).half()
    wt = torch.randn(cout, cin, *k3) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5
    cw = ops.pack_conv(wt, 0.1 * torch.randn(cout), geglu=geglu, device=dev)
    n_out = cout // 2 if geglu else cout
    r = None if res is None else (torch.randn(rows, n_out, device=dev) if res == "f32" else torch.randn(rows, n_out, device=dev).half())
    for i in range(3):
        if i == 2:
            print(name, file=sys.stderr, flush=True)
        os.environ["UAV_W4_TRACE_QUIET"] = "0"
        ops.conv_gemm(x, cw, n_img=n_img, t_len=t_len, hi=h, wi=w, residual=r, out_f32=out_f32)
    torch.cuda.synchronize()


if os.environ.get("UAV_TRACE_SMALL"):
    # Is the epilogue of the K = 512 linears bound by the CU (vector-memory transactions: a lane owns a ROW, so every 16-B piece of a
    # wave's load / store is its own cache line) or by the chip (all 256 CUs in their epilogues at once)?  The same tile on 64 / 128 /
    # 256 / 3200 workgroups (UAV_CONV_TILE=256 forces the big tile on the small grids): per-tile epilogue ticks that do not fall on
    # the small grids are CU-bound.
    for m in (8192, 16384, 32768, 409600):
        run(f"lin 512->512 M={m} ->f16", 1, 1, m, 1, 512, 512, (1, 1, 1))
        run(f"lin 512->512 M={m} res32->f32", 1, 1, m, 1, 512, 512, (1, 1, 1), res="f32", out_f32=True)
        run(f"lin 512->512 M={m} ->f32", 1, 1, m, 1, 512, 512, (1, 1, 1), out_f32=True)
    sys.exit(0)

run("lin 512->512 M=409600 ->f16", 1, 1, 409600, 1, 512, 512, (1, 1, 1))
run("lin 512->512 M=409600 res32->f32", 1, 1, 409600, 1, 512, 512, (1, 1, 1), res="f32", out_f32=True)
run("lin 512->4096 geglu M=409600", 1, 1, 409600, 1, 512, 4096, (1, 1, 1), geglu=True)
run("lin 1024->1024 M=25600 res32->f32", 1, 1, 25600, 1, 1024, 1024, (1, 1, 1), res="f32", out_f32=True)
run("lin 2048->512 M=409600 res32->f32", 1, 1, 409600, 1, 2048, 512, (1, 1, 1), res="f32", out_f32=True)
run("3x3 512->512 @16x160x160 ->f32", 16, 8, 160, 160, 512, 512, (1, 3, 3), out_f32=True)
run("t3 256->256 @16x320x320 res32->f32", 16, 8, 320, 320, 256, 256, (3, 1, 1), res="f32", out_f32=True)
