"""How much of a UNet forward / a whole clip is HOST time (Python + ctypes launch path)?  Prints the time the host needs to
ISSUE one UNetVideoModel.forward at the headline shape (call returns, no sync) next to the GPU time of the same forward, and
the per-level breakdown of issue vs GPU time (down / mid / up blocks), to see where the GPU can run dry."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))


def main():
    import bench
    from uav import ops
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev, 320, 320, text_encoder="standin")
    unet = pipe.unet
    g = torch.Generator(device=dev).manual_seed(1)
    lat = torch.randn(1, 4, 8, 320, 320, generator=g, device=dev).repeat(2, 1, 1, 1, 1)
    low = torch.randn(1, 3, 8, 320, 320, generator=g, device=dev).repeat(2, 1, 1, 1, 1)
    ehs = torch.randn(2, 77, 1024, generator=g, device=dev).half()
    cl = torch.tensor([120])
    with torch.no_grad():
        for _ in range(2):
            unet(lat, 925, low, encoder_hidden_states=ehs, class_labels=cl, cfg_shared_input=True)
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            unet(lat, 925, low, encoder_hidden_states=ehs, class_labels=cl, cfg_shared_input=True)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_total = time.perf_counter() - t0
            print(f"forward: host issue {t_issue * 1e3:7.1f} ms, until GPU done {t_total * 1e3:7.1f} ms", flush=True)
        # launches per forward
        ops.PROFILER.start()
        unet(lat, 925, low, encoder_hidden_states=ehs, class_labels=cl, cfg_shared_input=True)
        torch.cuda.synchronize()
        ops.PROFILER.stop()
        summ = ops.PROFILER.summary()
        n = sum(v["launches"] for v in summ.values())
        gpu = sum(v["seconds"] for v in summ.values())
        print(f"profiled launches per forward: {n}, summed kernel time {gpu * 1e3:.1f} ms")
        short = sum(1 for k, fl, nb, e0, e1 in ops.PROFILER.records if e0.elapsed_time(e1) < 0.03)
        print(f"launches shorter than 30 us: {short}")


if __name__ == "__main__":
    main()
