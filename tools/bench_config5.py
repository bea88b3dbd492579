"""BASELINE configs[4], ONE whole clip on one GPU: 32 frames 540x960 -> 2160x3840, the CLI's tiled branch
(`uav.tiling.upscale_tiled`, tile_size 256: 8 tiles of (320|348) x (320|384|384|256) px with their 64-px context), 6 temporal
windows (5 unique) x 30 DDIM steps per tile, `--use_video_vae` decoder (vae_video config: SFT conditioning on the LR frames,
3x3x3 convs), Wavelet colour fix against the bicubic-upsampled LR frames (inference_upscale_a_video.py:207-333).  The timed
region is the reference CLI's own (:205-206 ... :337-338 minus the `.cpu()` copy): tile loop + colour fix, inputs resident.

    python tools/bench_config5.py [--frames 32] [--height 540] [--width 960] [--ddim-steps 30] [--unet-stream f32|f16]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--ddim-steps", type=int, default=30)
    ap.add_argument("--tile-size", type=int, default=256)
    ap.add_argument("--unet-stream", choices=["f16", "f32"], default=None)
    ap.add_argument("--color-fix", choices=["Wavelet", "AdaIn", "None"], default="Wavelet")
    args = ap.parse_args()
    import bench
    from uav import configs, ops, tiling
    from models_video import color_correction as CC
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pipe = bench.build_pipeline(dev, args.height, args.width, vae_cfg=configs.VAE_VIDEO)
    pipe.vae.stream_dtype = torch.float32
    if args.unet_stream is not None:
        pipe.unet.stream_dtype = torch.float32 if args.unet_stream == "f32" else torch.float16
    clip = bench.synthetic_clip(args.frames, args.height, args.width, seed=0, dev=dev)
    tiles = tiling.tile_grid(args.height, args.width, args.tile_size)
    kw = dict(num_inference_steps=args.ddim_steps, guidance_scale=6.0, noise_level=120, negative_prompt="blur, worst quality",
              propagation_steps=[])
    # warm-up: one tile-sized 8-frame clip at 2 DDIM steps (weight packing, text encoder, allocator)
    pipe("best quality, extremely detailed", image=clip[:, :, :8, :320, :320].contiguous(), generator=torch.Generator().manual_seed(1),
         **dict(kw, num_inference_steps=2))
    torch.cuda.synchronize()
    ops.PROFILER.start(only={"conv_gemm"})
    t0 = time.perf_counter()
    gen = torch.Generator().manual_seed(10)
    out = tiling.upscale_tiled(pipe, "best quality, extremely detailed", clip, None, gen, tile_size=args.tile_size, **kw)
    torch.cuda.synchronize(); t_tiles = time.perf_counter() - t0
    frames_tchw = out[0].permute(1, 0, 2, 3).contiguous()
    if args.color_fix != "None":
        style = CC.upsample_bicubic4(clip[0].permute(1, 0, 2, 3).contiguous()) if hasattr(CC, "upsample_bicubic4") else \
            torch.nn.functional.interpolate(clip[0].permute(1, 0, 2, 3).contiguous(), scale_factor=4, mode="bicubic")
        frames_tchw = CC.wavelet_reconstruction(frames_tchw, style) if args.color_fix == "Wavelet" else \
            CC.adaptive_instance_normalization(frames_tchw, style)
    torch.cuda.synchronize(); elapsed = time.perf_counter() - t0
    ops.PROFILER.stop()
    summ = ops.PROFILER.summary()["conv_gemm"]
    assert frames_tchw.shape == (args.frames, 3, 4 * args.height, 4 * args.width) and bool(torch.isfinite(frames_tchw).all())
    print(json.dumps({
        "metric": "upscaled frames/sec, BASELINE configs[4] (one clip, one GPU)", "value": args.frames / elapsed, "unit": "frames/s",
        "seconds_per_clip": elapsed, "seconds_tile_loop": t_tiles, "seconds_color_fix": elapsed - t_tiles, "n_gpus": 1,
        "config": {"workload": f"configs[4], ONE clip: {args.frames}-frame {args.height}x{args.width}->{4 * args.height}x{4 * args.width}, "
                               f"CLI tile loop (tile_size {args.tile_size}: {len(tiles)} tiles {sorted({(t.src[1] - t.src[0], t.src[3] - t.src[2]) for t in tiles})}), "
                               f"{args.ddim_steps} DDIM steps, guidance 6, sliding 8-frame windows (5 unique per step for T = 32), vae_video decoder (fp32 "
                               f"stream), UNet stream {'fp32' if pipe.unet.stream_f32() else 'fp16'}, {args.color_fix} colour fix; random-init weights, synthetic clip"},
        "conv_gemm": {"launches": summ["launches"], "seconds": summ["seconds"], "tflops": summ["flops"] / summ["seconds"] / 1e12,
                      "frac_of_2.5PF": summ["flops"] / summ["seconds"] / 2.5e15},
        "algorithmic_pflop_survey_unique_windows": 232.7 + 16.7 if (args.frames, args.height, args.width, args.ddim_steps) == (32, 540, 960, 30) else None,
        "data": "synthetic", "dtype": "f16"}), flush=True)


if __name__ == "__main__":
    main()
