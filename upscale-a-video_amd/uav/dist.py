"""One-process-per-GPU helpers (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU).

The hot path shards by CLIP (and by spatial tile of a clip): units are independent, so the data path
needs no collective — ranks only meet at the timing barrier and at the optional gather
of results.  Everything here is backend-agnostic so the N>1 logic is covered by world_size-2 gloo tests
on CPU.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the default process group from torchrun's environment (no-op for world size 1)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (see task environment notes)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend)
    return world, rank, local


def shard(items, rank, world):
    """Round-robin assignment of independent work units (clips / tiles) to ranks: unit i -> rank i % world."""
    return [it for i, it in enumerate(items) if i % world == rank]


def sharded_map(items, fn, group=None, like=None):
    """Evaluate `fn(item) -> tensor` for every item with the items dealt round-robin over the ranks, then
    all-gather: every rank returns the full list, in item order, bit-identical to the unsharded evaluation.

    This is how ONE long clip is spread over GPUs (BASELINE config 4): the temporal windows of a DDIM step are
    independent UNet evaluations — the only coupling is the epsilon blend on the overlap frames, which
    every rank then replays on the gathered outputs — and so are the 3-frame VAE decode chunks.  All results of one
    call must share shape and dtype (windows are all 8 frames; the ragged last decode chunk is padded by the
    caller).  One all_gather of (ceil(n/world), *shape) per call; RCCL over xGMI when the backend is nccl.
    `like` = (shape, dtype, device) of one result: lets a rank that owns no item (more ranks than items) take part
    without the object broadcast that is otherwise needed to learn the shape."""
    items = list(items)
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or len(items) <= 1:
        return [fn(it) for it in items]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = [fn(it) for it in items[rank::world]]
    per_rank = -(-len(items) // world)
    if mine:
        shape, dtype, device = tuple(mine[0].shape), mine[0].dtype, mine[0].device
    elif like is not None:
        shape, dtype, device = tuple(like[0]), like[1], like[2]
    if len(items) < world and like is None:
        # some rank owns nothing and has no hint: rank 0 (which always owns item 0) tells everybody the shape
        meta = [(shape, dtype) if mine else None]
        dist.broadcast_object_list(meta, src=0, group=group)
        shape, dtype = meta[0]
        device = mine[0].device if mine else _default_device()
    buf = torch.zeros((per_rank,) + shape, dtype=dtype, device=device)
    for i, t in enumerate(mine):
        buf[i] = t
    gathered = _all_gather(buf, world, group)
    return [gathered[i % world][i // world] for i in range(len(items))]


def _all_gather(buf, world, group=None):
    """all_gather of one tensor per rank.  RCCL ("nccl") gathers device tensors in place over xGMI; the gloo backend (CPU tests,
    and the 2-ranks-on-ONE-GPU check of tests/test_multigpu_gpu.py, where RCCL refuses two ranks on a device) is staged
    through host memory — bit-exact either way."""
    if dist.get_backend(group) == "gloo" and buf.is_cuda:
        host = buf.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return [p_.to(buf.device) for p_ in parts]
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return parts


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def barrier(device=None):
    if dist.is_initialized():
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (the timed region's duration)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_to_rank0(obj):
    """Gather one picklable object per rank on rank 0 (None elsewhere)."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
