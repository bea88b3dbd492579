"""Calibration of the box: what does a PLAIN fp16 GEMM reach here, on the data the conv kernel sees?
  (a) torch.matmul on the shape M = 409600, N = 512, K = 4608 (the vendor's hipBLASLt / Tensile assembly kernel — test
      infrastructure only, never the product path), operands N(0,1) -> SiLU like the conv inputs behind a GroupNorm;
  (b) uav_conv_gemm_f16 in 1x1 mode on the same shape (no gather arithmetic: row m is pixel m);
  (c) the 3x3 512 -> 512 @16x160x160 implicit GEMM (same M, N, K = 9 x 512; nine shifted gathers);
  (d) the same three on ZERO operands (DVFS give-back: guide rule 25).
One JSON line per arm; run under `rocprofv3 --pmc` for the SQ counters (tools/run.sh calib_pmc)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 409600, 512, 4608
ARMS = sys.argv[1:] or ["blas", "conv1x1", "conv3x3"]
ITERS = int(os.environ.get("UAV_CALIB_ITERS", "8"))


def timeit(fn, iters=ITERS, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return min(ts), sorted(ts)[1]


def operands(zero):
    if zero:
        return torch.zeros(M, K, device=dev, dtype=torch.float16), torch.zeros(N, K, device=dev, dtype=torch.float16)
    x = torch.nn.functional.silu(torch.randn(M, K, device=dev)).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    return x, w


def main():
    fl = 2.0 * M * N * K
    for zero in (False, True):
        x, w = operands(zero)
        tag = "zeros" if zero else "silu(randn)"
        if "blas" in ARMS:
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            wt = w.t()
            mn, med = timeit(lambda: torch.matmul(x, wt, out=out))
            print(json.dumps({"arm": "a_plain_gemm_hipblaslt", "data": tag, "ms_min": mn, "ms_median": med, "tflops": fl / med / 1e9}), flush=True)
        if "conv1x1" in ARMS:
            cw = ops.pack_conv(w.float().cpu().reshape(N, K, 1, 1, 1), torch.zeros(N), device=dev)
            n_img, hi = ops._factor_rows(M)
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            mn, med = timeit(lambda: ops.conv_gemm(x, cw, n_img=n_img, t_len=1, hi=hi, wi=1, out=out))
            print(json.dumps({"arm": "b_conv_kernel_1x1_K4608", "data": tag, "ms_min": mn, "ms_median": med, "tflops": fl / med / 1e9}), flush=True)
        if "conv3x3" in ARMS:
            x3 = x[:, :512].contiguous()
            w3 = w.float().cpu().reshape(N, 9, 512).permute(0, 2, 1).reshape(N, 512, 1, 3, 3).contiguous()
            cw = ops.pack_conv(w3, torch.zeros(N), device=dev)
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            mn, med = timeit(lambda: ops.conv_gemm(x3, cw, n_img=16, t_len=8, hi=160, wi=160, out=out))
            print(json.dumps({"arm": "c_conv_kernel_3x3_512", "data": tag, "ms_min": mn, "ms_median": med, "tflops": fl / med / 1e9}), flush=True)
        del x, w


if __name__ == "__main__":
    main()
