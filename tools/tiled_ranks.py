"""One clip upscaled through the CLI's tile loop (uav.tiling.upscale_tiled) on 1 or N ranks; rank 0 prints a JSON line with the
sha256 of the stitched output.  The tiles are dealt over the ranks, every rank replays the shared generator's draws up to its
tiles, the disjoint output boxes are merged by one all-reduce: the digest must not depend on the world size
(tests/test_multigpu_gpu.py).  UAV_BENCH_SAME_GPU=1: all ranks on GPU 0, gloo transport (1-GPU boxes).

    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 2 tools/tiled_ranks.py
"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))


def main():
    import bench
    from uav import tiling
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    same = os.environ.get("UAV_BENCH_SAME_GPU") == "1"
    gpu = 0 if same else local
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if same else "nccl", **({} if same else {"device_id": dev}))
    pipe = bench.build_pipeline(dev, 64, 64, text_encoder="standin")
    clip = bench.synthetic_clip(2, 72, 200, seed=3, dev=dev)                     # 72 x 200 at tile 64 -> 1 x 3 tiles, H not a multiple of 8
    out = tiling.upscale_tiled(pipe, "best quality", clip, None, torch.Generator().manual_seed(10), tile_size=64,
                               num_inference_steps=2, guidance_scale=6.0, noise_level=120, negative_prompt="blur", propagation_steps=[])
    torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps({"world": world, "tiles": len(tiling.tile_grid(72, 200, 64)), "shape": list(out.shape),
                          "output_sha256": hashlib.sha256(out.float().cpu().numpy().tobytes()).hexdigest()}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
