"""The VAE's d = 512 single-head attention at the headline size (L = 320 x 320 = 102 400 keys per frame): the ring kernel on pre-packed
K / V^T (csrc/attn512x.hip, default) or attn512w_kernel (UAV_ATTN512X=0), one JSON line.  usage: python tools/bench_attn512.py [frames]"""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")
bq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, d = 102400, 512
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(bq * L, 3 * d, generator=g) * 0.9).half().to(dev)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
f = lambda: ops.attention(q, k, v, bq=bq, lq=L, lk=L, heads=1, head_dim=d, q_stride=3 * d, k_stride=3 * d, v_stride=3 * d)
f(); torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("UAV_XA_ROUNDS", "3"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = statistics.median(ts)
print(json.dumps({"kernel": "attn512x (ring, packed K | V^T)" if ops.ATTN512X else "attn512w", "frames": bq, "L": L, "ms": round(ms, 3),
                  "tflops": round(4.0 * bq * L * L * d / ms / 1e9, 1)}), flush=True)
