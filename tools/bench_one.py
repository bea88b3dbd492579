"""Run ONE conv shape a few times (for rocprofv3 --pmc passes).  usage: bench_one.py <case> [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops
dev = torch.device("cuda:0")
CASES = {"c512_320": (16, 8, 320, 320, 512, 512, (1, 3, 3)), "lin512": (640, 1, 640, 1, 512, 512, (1, 1, 1)),
         "c256_320": (16, 8, 320, 320, 256, 256, (1, 3, 3)), "c1024_40": (16, 8, 40, 40, 1024, 1024, (1, 3, 3))}
n_img, t_len, h, w, cin, cout, k3 = CASES[sys.argv[1]]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = torch.randn(n_img * h * w, cin, device=dev).half()
cw = ops.pack_conv(torch.randn(cout, cin, *k3) * (cin * k3[0] * k3[1] * k3[2]) ** -0.5, torch.zeros(cout), device=dev)
for _ in range(iters):
    y = ops.conv_gemm(x, cw, n_img=n_img, t_len=t_len, hi=h, wi=w)
torch.cuda.synchronize()
