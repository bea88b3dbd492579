"""What the vendor's plain fp16 GEMM (torch.matmul -> hipBLASLt; calibration only, never the product path) reaches on the
GEMM shapes of the UNet's linears / 1x1 convs and on the implicit-GEMM shapes of its 3x3 convs (K = taps x C_in), on
SiLU(N(0,1)) operands.  One JSON line per shape: compare with tools/bench_shortk.py / bench_epilogue.py on the same box."""
import json
import sys

import torch

dev = torch.device("cuda:0")
SHAPES = [(409600, 512, 512), (409600, 1536, 512), (409600, 4096, 512), (409600, 512, 2048), (409600, 512, 1024),
          (102400, 512, 512), (25600, 1024, 1024), (25600, 3072, 1024), (1638400, 256, 256), (1638400, 256, 512),
          (409600, 512, 4608), (1638400, 256, 2304), (25600, 1024, 9216), (409600, 512, 1536)]


def timeit(fn, iters=10, warm=3):
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
    return sorted(ts)[1]


for m, n, k in SHAPES:
    x = torch.nn.functional.silu(torch.randn(m, k, device=dev)).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    out = torch.empty(m, n, device=dev, dtype=torch.float16)
    wt = w.t()
    ms = timeit(lambda: torch.matmul(x, wt, out=out))
    print(json.dumps({"M": m, "N": n, "K": k, "ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1),
                      "GBps_min_traffic": round((2.0 * m * k + 2.0 * m * n) / ms / 1e6, 0)}), flush=True)
    del x, w, out
