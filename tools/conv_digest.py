"""Prints a digest of conv_gemm outputs on fixed seeded inputs (GPU box only): used to check that two builds / kernel
variants (UAV_CONV_DMAV, UAV_CONV_TILE, ...) are BIT-identical — same tile walk, same accumulation order."""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")
CASES = [("3x3_512_160", 16, 8, 160, 160, 512, 512, (1, 3, 3), 1, False), ("t5_256_80", 16, 8, 80, 80, 256, 256, (5, 1, 1), 1, False),
         ("lin_512_4096", 160, 1, 640, 1, 512, 4096, (1, 1, 1), 1, False), ("up_256_40", 6, 3, 40, 40, 256, 256, (1, 3, 3), 1, True),
         ("s2_256_81x79", 4, 2, 81, 79, 256, 512, (1, 3, 3), 2, False), ("3x3x3_256_33", 6, 3, 33, 35, 256, 256, (3, 3, 3), 1, False),
         ("cat_1536_40", 16, 8, 40, 40, 1536, 512, (1, 1, 1), 1, False)]
out = {}
for name, n_img, t_len, h, w, cin, cout, k3, stride, ups in CASES:
    g = torch.Generator().manual_seed(len(name) * 131 + cin)
    x = torch.randn(n_img * h * w, cin, generator=g).half().to(dev)
    wt = (torch.randn(cout, cin, *k3, generator=g) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5)
    cw = ops.pack_conv(wt, torch.randn(cout, generator=g), device=dev, geglu=name.startswith("lin_512_4096"))
    y = ops.conv_gemm(x, cw, n_img=n_img, t_len=t_len, hi=h, wi=w, stride=stride, upsample=ups)
    torch.cuda.synchronize()
    out[name] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(out))
