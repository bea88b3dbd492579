"""Spatial tiling of a clip — the reference CLI's tile loop as a schedulable work list.

`inference_upscale_a_video.py:201-304` upscales frames of 384x384 pixels or more in overlapping tiles: `tile_size`
squares, 64 px of context on every side that has any, one `pipeline(...)` call per tile in row-major order with ONE
shared generator, and the un-padded centre of each 4x result pasted into the output.  Here the same arithmetic produces
a list of `Tile` records, and `upscale_tiled` runs them

  * serially (world size 1): identical to the CLI loop, call for call;
  * dealt round-robin over the ranks of a process group (BASELINE config 5: data-parallel over tiles and clips): tiles
    are independent except for the shared generator, so every rank replays the generator's draw sequence up to its
    tile (two `randn` per tile, in the pipeline's order: LR noise, then latents — `pipeline_upscale_a_video.py:546-551,
    573`) and the result is bit-identical to the serial loop.  The disjoint output boxes are merged by one all-reduce
    of the output canvas (adding zeros is exact).

Only index arithmetic and scheduling live here; the per-tile work is `VideoUpscalePipeline.__call__`.
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist

TILE_OVERLAP = 64            # `tile_overlap_height = tile_overlap_width = 64 # should be >= 64` (inference:210)
SCALE = 4


@dataclass(frozen=True)
class Tile:
    """All boxes are (y0, y1, x0, x1).  `src`: padded input box on the LR frame; `dst`: box on the 4x output canvas;
    `crop`: the part of the 4x tile result that lands in `dst`."""
    index: int
    src: tuple
    dst: tuple
    crop: tuple


def needs_tiling(h: int, w: int, perform_tile: bool = False) -> bool:
    """`if h * w >= 384*384: args.perform_tile = True` (inference:201-202)."""
    return perform_tile or h * w >= 384 * 384


def tile_grid(h: int, w: int, tile_size: int = 256, overlap: int = TILE_OVERLAP, scale: int = SCALE) -> List[Tile]:
    """Tile boxes in the CLI's visiting order (inference:209-303), including its end-of-row rule: when the last
    tile would start inside the previous tile's right/bottom context (`(tiles-1)*tile + overlap >= size`), that tile is
    dropped and the previous one is extended to the frame edge on the OUTPUT side (`rm_end_pad_* = False`)."""
    tiles_x, tiles_y = math.ceil(w / tile_size), math.ceil(h / tile_size)
    rm_end_pad_w = rm_end_pad_h = True
    if (tiles_x - 1) * tile_size + overlap >= w:
        tiles_x, rm_end_pad_w = tiles_x - 1, False
    if (tiles_y - 1) * tile_size + overlap >= h:
        tiles_y, rm_end_pad_h = tiles_y - 1, False
    out_h, out_w = h * scale, w * scale
    tiles = []
    for y in range(tiles_y):
        for x in range(tiles_x):
            x0, y0 = x * tile_size, y * tile_size
            x1, y1 = min(x0 + tile_size, w), min(y0 + tile_size, h)
            x0p, x1p = max(x0 - overlap, 0), min(x1 + overlap, w)
            y0p, y1p = max(y0 - overlap, 0), min(y1 + overlap, h)
            last_x = x == tiles_x - 1 and not rm_end_pad_w
            last_y = y == tiles_y - 1 and not rm_end_pad_h
            ox0, oy0 = x0 * scale, y0 * scale
            ox1 = out_w if last_x else x1 * scale
            oy1 = out_h if last_y else y1 * scale
            cx0, cy0 = (x0 - x0p) * scale, (y0 - y0p) * scale
            cx1 = cx0 + (out_w - ox0 if last_x else (x1 - x0) * scale)
            cy1 = cy0 + (out_h - oy0 if last_y else (y1 - y0) * scale)
            tiles.append(Tile(len(tiles), (y0p, y1p, x0p, x1p), (oy0, oy1, ox0, ox1), (cy0, cy1, cx0, cx1)))
    return tiles


def generator_states(generator: torch.Generator, tiles: List[Tile], frames: int, lr_channels: int, latent_channels: int,
                     draw_dtype: torch.dtype, device) -> list:
    """State of the shared generator at the start of every tile's pipeline call, obtained by replaying the two draws
    each earlier tile makes (LR noise of the padded tile, then the latents).  The generator is left in the state the
    serial loop would leave it in (after the last tile)."""
    from models_video.pipeline_upscale_a_video import randn_tensor
    states = []
    for tl in tiles:
        states.append(generator.get_state())
        th, tw = tl.src[1] - tl.src[0], tl.src[3] - tl.src[2]
        randn_tensor((1, lr_channels, frames, th, tw), generator=generator, device=device, dtype=draw_dtype)
        randn_tensor((1, latent_channels, frames, th, tw), generator=generator, device=device, dtype=draw_dtype)
    return states


def upscale_tiled(pipeline, prompt, vframes: torch.Tensor, flows_bi: Optional[list], generator: torch.Generator, *,
                  tile_size: int = 256, group=None, **pipeline_kwargs) -> torch.Tensor:
    """The CLI's tiled branch.  vframes (1,C,T,h,w) in [-1,1]; returns (1,C,T,4h,4w) on every rank.
    `pipeline_kwargs`: num_inference_steps, guidance_scale, noise_level, negative_prompt, propagation_steps."""
    b, c, t, h, w = vframes.shape
    tiles = tile_grid(h, w, tile_size)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    out = vframes.new_zeros((b, c, t, h * SCALE, w * SCALE))                    # "start with black image"
    if world > 1:
        draw_dtype = getattr(getattr(pipeline, "text_encoder", None), "dtype", torch.float32)
        latent_c = pipeline.vae.config.latent_channels
        states = generator_states(generator, tiles, t, c, latent_c, draw_dtype, vframes.device)
        final_state = generator.get_state()
    for tl in tiles:
        if tl.index % world != rank:
            continue
        y0, y1, x0, x1 = tl.src
        if world > 1:
            generator.set_state(states[tl.index])
        tile_flows = None if flows_bi is None else [f[:, :, :, y0:y1, x0:x1] for f in flows_bi]
        res = pipeline(prompt, image=vframes[:, :, :, y0:y1, x0:x1], flows_bi=tile_flows, generator=generator,
                       **pipeline_kwargs).images
        dy0, dy1, dx0, dx1 = tl.dst
        cy0, cy1, cx0, cx1 = tl.crop
        out[:, :, :, dy0:dy1, dx0:dx1] = res[:, :, :, cy0:cy1, cx0:cx1]
    if world > 1:
        generator.set_state(final_state)
        if dist.get_backend(group) == "gloo" and out.is_cuda:                   # gloo (tests): staged through host memory
            host = out.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            out.copy_(host)
        else:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)           # disjoint boxes: x + 0 is exact
    return out
