#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: BASELINE.json's metric
"upscaled frames/sec (4x, 30 DDIM steps, 8-frame 320p clip)".

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one synthetic 8-frame 320x320 clip:
`VideoUpscalePipeline.__call__` with 30 DDIM steps, guidance 6, noise level 120, no propagation,
vae_3d decode to 1280x1280 (BASELINE.json configs[1]).  Weights are random-init at the released
architecture (691 M-param UNet + 55 M-param VAE); the LR clip is already resident in HBM when the
timed region starts.  With N > 1 every rank (one process per GPU) upscales its own clips: clips are
independent, there is no data-path collective -> weak scaling; the timed region is
bracketed by barrier + synchronize on both sides and the MAX over ranks is reported.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     dominant kernel (implicit-GEMM conv/linear): algorithmic FLOP / HIP-event time per
               launch over the timed region vs the dense fp16 MFMA peak (2.5 PFLOP/s)
  cpu_baseline the oracle (CPU fp32 restatement of the reference path) timed on this host's cores on
               a bounded sample (N=1 only), extrapolated by FLOP — a reported baseline, not the target
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))

METRIC = "upscaled frames/sec (4x, 30 DDIM steps, 8-frame 320p clip)"
PEAK_TFLOPS_F16 = 2500.0        # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0          # HBM3E spec peak (6.3 TB/s measured achievable), same guide


def build_text_encoder(dev, dim, kind):
    """`clip`: the ViT-H/14 text tower of the released pipeline (24 layers, width 1024, 16 heads, 354 M parameters, random
    init) on the HIP kernels (uav/clip_text.py); `standin`: the deterministic embedding table of SURVEY.md §8 row a19."""
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    tok = StandInTokenizer()
    if kind == "standin":
        return StandInTextEncoder(tok, dim), tok
    from uav.clip_text import UavCLIPTextModel
    torch.manual_seed(99)
    te = UavCLIPTextModel(vocab_size=49408, hidden_size=dim, intermediate_size=4 * dim, num_hidden_layers=24,
                          num_attention_heads=dim // 64, max_position_embeddings=77, hidden_act="gelu").half().to(dev).eval()
    return te, tok


def build_pipeline(dev, height, width, unet_cfg=None, vae_cfg=None, text_encoder="clip"):
    from uav import configs, init_weights
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from models_video.unet_video import UNetVideoModel
    unet_cfg = unet_cfg or configs.UNET_VIDEO
    vae_cfg = vae_cfg or configs.VAE_3D
    unet = UNetVideoModel.from_config(dict(unet_cfg)).half().to(dev).eval()
    init_weights.random_init_(unet, seed=1234)
    # fp32 parameters like the reference CLI builds it (inference_upscale_a_video.py:103-110: only the UNet is `.half()`): the
    # decoder then runs in the pipeline's DEFAULT mode for such a VAE (fp32 residual stream, fp16 MFMA operands)
    vae = AutoencoderKLVideo.from_config(dict(vae_cfg)).to(dev).eval()
    init_weights.random_init_(vae, seed=4321)
    te, tok = build_text_encoder(dev, unet_cfg["cross_attention_dim"], text_encoder)
    pipe = VideoUpscalePipeline(text_encoder=te, tokenizer=tok,
                                low_res_scheduler=DDPMScheduler(**configs.LOW_RES_DDPM),
                                scheduler=DDIMScheduler(**configs.DDIM), vae=vae, unet=unet, propagator=None).to(dev)
    return pipe


def predicted_scaling(args, world, value):
    """What the driver's 1 -> 8 GPU run of THIS command should show, printed by the command itself so that
    the SCALE record is held against a figure from the same line.  Clip-parallel default (weak scaling): no data-path
    collective, one barrier + one MAX-reduce per timed region -> N x the per-GPU rate, efficiency >= 0.97 (shared: host cores
    for ~40 k launches per clip and rank, the node's power envelope).  --shard-windows (strong scaling, ONE clip): the unique
    temporal windows of a DDIM step (x 2 guidance branches with --shard-cfg, +2.7 % FLOP) and the 3-frame decode chunks are
    the units; the critical path per step is ceil(units / N) unit times, the all-gather (131 MB per step, ~1 ms over xGMI)
    and the replicated CFG / DDIM / propagation are the serial remainder (~1 %)."""
    import math
    per_gpu = value / (1 if args.shard_windows else world)
    ns = (1, 2, 4, 8)
    if not args.shard_windows:
        return {"mode": "clip-parallel (weak scaling)", "basis_frames_per_s_per_gpu": per_gpu, "assumed_efficiency": 0.97,
                "frames_per_s": {str(n): per_gpu * n * (1.0 if n == 1 else 0.97) for n in ns}}
    from models_video.pipeline_upscale_a_video import window_schedule
    wins = window_schedule(args.frames)
    n_win = len(set(wins))
    units = n_win * (2 if args.shard_cfg else 1)
    chunks = math.ceil(args.frames / 3)
    # one unit = one window evaluation (or one guidance branch of it); decode chunks cost ~1.37 window units of a B2 window
    # TFLOP of an 8-frame 320x320 window / a 3-frame decode chunk: only their RATIO enters the prediction, and it is the
    # ratio of the default shape — `unit_costs` on the line says so when another --height / --width runs
    win_flop, chunk_flop = 176.99 * (1.027 / 2 if args.shard_cfg else 1.0), 299.05 / 3
    def t(n):
        return args.ddim_steps * math.ceil(units / n) * win_flop + math.ceil(chunks / n) * chunk_flop
    base = value if world == 1 else None
    return {"mode": "one clip, units dealt over the ranks (strong scaling)", "units_per_step": units, "decode_chunks": chunks,
            "unit_costs": "window : chunk FLOP ratio of the default 320x320 shape" + ("" if (getattr(args, "height", 320), getattr(args, "width", 320)) == (320, 320) else " (another shape runs: approximate)"),
            "speedup_vs_1_gpu": {str(n): t(1) / t(n) for n in ns},
            "frames_per_s": ({str(n): base * t(1) / t(n) * (1.0 if n == 1 else 0.99) for n in ns} if base else None)}


def synthetic_clip(frames, height, width, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((1, 3, frames, height + 4, width + 4), generator=g) * 2 - 1
    x = torch.nn.functional.avg_pool3d(x, (1, 5, 5), stride=1)           # low-pass: trackable structure
    return (x / x.abs().max()).contiguous().to(dev)


def cpu_baseline(sample_hw=64, frames=8, max_threads=32):
    """Oracle (CPU fp32 restatement of the reference path) on the full-width model, bounded sample:
    ONE UNet forward (CFG batch 2, 8 frames) + ONE 3-frame VAE decode chunk at sample_hw x sample_hw,
    extrapolated to config 2 by the analytic FLOP model."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import synth
    import uav_oracle as O
    from uav import configs
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.unet_video import UNetVideoModel
    cores = min(os.cpu_count() or 1, max_threads)      # more threads than that slow ATen's CPU convs down
    torch.set_num_threads(cores)
    shapes = {k: tuple(v.shape) for k, v in UNetVideoModel.from_config(dict(configs.UNET_VIDEO)).state_dict().items()}
    g = torch.Generator().manual_seed(0)
    usd = {}
    for k, shp in shapes.items():
        if k.endswith("freqs"):
            usd[k] = synth.synth_tensor(k, shp)
        elif len(shp) == 1 and "norm" in k and k.endswith("weight"):
            usd[k] = torch.ones(shp)
        else:
            fan = 1
            for s in shp[1:]:
                fan *= s
            usd[k] = torch.randn(shp, generator=g) * (fan ** -0.5 if len(shp) > 1 else 0.1)
    vshapes = {k: tuple(v.shape) for k, v in AutoencoderKLVideo.from_config(dict(configs.VAE_3D)).state_dict().items()}
    vsd = {}
    for k, shp in vshapes.items():
        fan = 1
        for s in shp[1:]:
            fan *= s
        vsd[k] = torch.ones(shp) if (len(shp) == 1 and "norm" in k and k.endswith("weight")) else \
            torch.randn(shp, generator=g) * (fan ** -0.5 if len(shp) > 1 else 0.1)
    h = w = sample_hw
    sample = torch.randn(2, 4, frames, h, w, generator=g); low = torch.randn(2, 3, frames, h, w, generator=g)
    ehs = torch.randn(2, 77, 1024, generator=g)
    with torch.no_grad():
        t0 = time.time()
        O.unet_forward(usd, configs.UNET_VIDEO, sample, 925, low, ehs, torch.tensor([120]))
        t_unet = time.time() - t0
        t0 = time.time()
        O.vae_decode(vsd, configs.VAE_3D, torch.randn(1, 4, 3, h, w, generator=g), None, 1.0)
        t_vae = time.time() - t0
    # FLOP of the sample (conv/linear parts scale with pixels; attention cores with pixels^2 — small at 64x64)
    scale = (sample_hw / 320.0) ** 2
    tf_unet = configs.TFLOP_UNET_320 * scale
    tf_vae_conv = (configs.TFLOP_VAE3D_320 - 171.80) * scale * 3 / 8
    tf_vae_attn = 171.80 * scale * scale * 3 / 8
    rate = (tf_unet + tf_vae_conv + tf_vae_attn) / (t_unet + t_vae)                 # TFLOP/s on this host
    clip_seconds = configs.TFLOP_CLIP_C2 / rate
    return {"value": 8.0 / clip_seconds, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU fp32 restatement), full-width model: 1 UNet forward (B2xT{frames}x{h}x{w}, {t_unet:.1f}s) + "
                      f"1 three-frame VAE decode chunk ({h}x{w}, {t_vae:.1f}s) = {rate:.3f} TFLOP/s, extrapolated by "
                      f"FLOP to 5608.7 TFLOP/clip"}


def self_launch(n):
    """Re-exec this script as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same args>`."""
    import subprocess
    have = torch.cuda.device_count()
    if os.environ.get("UAV_BENCH_SAME_GPU") == "1" and have >= 1:
        have = n                        # test mode: all ranks on GPU 0, gloo transport (RCCL refuses two ranks per device)
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # --standalone: the launcher binds its own free rendezvous port (no bind-then-close race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--ddim-steps", type=int, default=30)
    ap.add_argument("--propagation", action="store_true",
                    help="BASELINE configs[2]: RAFT flows (computed before the timed region, like the reference CLI "
                         "inference:191 vs :205) + latent propagation at DDIM steps 24,26,28")
    ap.add_argument("--no-cfg-share", action="store_true",
                    help="A/B switch: run the text-independent UNet head for both guidance branches like the reference")
    ap.add_argument("--shard-windows", action="store_true",
                    help="BASELINE configs[3]: ONE long clip (use --frames 32) whose temporal windows and decode chunks are "
                         "dealt over the ranks (strong scaling); every rank feeds the same clip and seed")
    ap.add_argument("--shard-cfg", action="store_true",
                    help="with --shard-windows: split every UNet evaluation into its two guidance branches, i.e. deal (window x "
                         "branch) units over the ranks (10 per DDIM step for 32 frames, 2 for one 8-frame clip); bit-identical for "
                         "every world size")
    ap.add_argument("--digest", action="store_true",
                    help="add config.output_sha256 (rank 0's last output tensor) to the JSON line: the sharded modes promise the "
                         "same bits for every world size (tests/test_multigpu_gpu.py)")
    ap.add_argument("--clips-per-step", type=int, default=1,
                    help="clips upscaled CONCURRENTLY per GPU in one step, one HIP stream (and host thread) each: the "
                         "HBM-bound kernels of one clip run beside the MFMA-bound kernels of the other (serving mode); "
                         "per-launch HIP events are switched off because launches overlap")
    ap.add_argument("--overlap-streams", type=int, default=None,
                    help="a clip LONGER than 8 frames: its temporal windows and decode chunks issued on this many HIP streams from one "
                         "host thread (uav/streams.py; default: the pipeline's, 2; 0 = serial); same bits as serial.  Launches then "
                         "overlap, so `roofline` is measured in one extra SERIAL step.  The 8-frame headline clip has one window and "
                         "runs serially either way (unless --overlap-split-cfg)")
    ap.add_argument("--overlap-split-cfg", action="store_true",
                    help="with --overlap-streams: also split a window into its two guidance branches (batch-1 units, CFG-shared "
                         "head given up) when there are fewer windows than streams — measured slower on the 8-frame clip")
    ap.add_argument("--video-vae", action="store_true",
                    help="BASELINE configs[4] building block: the SFT-conditioned video VAE (vae_video config, 3x3x3 convs) instead "
                         "of vae_3d; use with --height 348 --width 384 for one CLI tile of a 540p frame")
    ap.add_argument("--vae-fp16", action="store_true",
                    help="A/B switch: all-fp16 VAE decoder rows (round-1 behaviour, ~1e-3 rel-L2 vs the fp32 reference decode) "
                         "instead of the default fp32 residual stream (~6e-4); the mode timed is named in config.workload")
    ap.add_argument("--unet-stream", choices=["f16", "f32"], default=None,
                    help="residual-stream precision of the UNet (UNetVideoModel.stream_dtype): f16 = every stored tensor fp16, the "
                         "arithmetic of the reference's `.half()` UNet; f32 = fp32 rows, fp32 latents between DDIM steps, fp16 MFMA "
                         "operands.  Default: the engine's default (models_video/unet_video.py DEFAULT_STREAM); named in config.workload")
    ap.add_argument("--precision", choices=["default", "high"], default=None,
                    help="UNetVideoModel.precision: 'high' also reads block tails as fp16 hi|lo pairs and keeps the ResNet branch tensor "
                         "fp32 (1.10e-3 instead of 1.23e-3 on unclamped pixels, +5.7 %% per clip); named in config.workload")
    ap.add_argument("--text-encoder", choices=["clip", "standin"], default="clip",
                    help="clip: ViT-H/14 text tower (random init) on the HIP kernels, run once per distinct prompt pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-throughput-mode", action="store_true",
                    help="skip the extra two-clips-per-GPU measurement (`throughput_mode` object) that follows the serial timed region")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--timed-kernel-events", action="store_true",
                    help="HIP events around the dominant kernel (conv_gemm) INSIDE the timed region, as rounds 1-5 measured `roofline`.  Default "
                         "since round 6: the timed region carries no events (same-box A/B, twice interleaved: 1.1417 with / 1.1489 frames/s "
                         "without the ~16 k event records per clip, +0.63 %%, profiles/r06_event_overhead_ab_*.jsonl) and `roofline` + the "
                         "per-kernel table come from one extra fully instrumented clip right behind it, same process, same box")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="HIP events around EVERY kernel family inside the timed region (costs ~2 %%: 50 k event records of ~3 us "
                         "of GPU time per clip).  Default: only the dominant kernel (conv_gemm, what `roofline` needs) is timed there "
                         "and the per-kernel table comes from one extra, untimed, fully instrumented step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (RCCL);
        # rank 0 of the child job prints the JSON line, which passes through unchanged.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    same_gpu = os.environ.get("UAV_BENCH_SAME_GPU") == "1"     # TEST MODE (tests/test_multigpu_gpu.py on a 1-GPU box): every rank on GPU 0
    gpu_index = 0 if same_gpu else local_rank
    if gpu_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {gpu_index} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_gpu:
            dist.init_process_group("gloo")                    # transport through host memory; the kernels are the real ones
        else:
            dist.init_process_group("nccl", device_id=dev)    # RCCL on ROCm; one process per GPU
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL reports {dist.get_world_size()} ranks, expected {args.gpus}")

    from uav import _lib, ops
    lib = _lib.load()
    if lib.uav_device_check(gpu_index, None) != 0:
        raise SystemExit("bench.py needs an MI355X (gfx950)")

    from uav import configs as _cfg
    pipe = build_pipeline(dev, args.height, args.width, text_encoder=args.text_encoder,
                          vae_cfg=_cfg.VAE_VIDEO if args.video_vae else None)
    if args.vae_fp16:
        pipe.vae = pipe.vae.half()              # what `pipeline.vae.half()` users get: all-fp16 decoder rows
    if args.unet_stream is not None:
        pipe.unet.stream_dtype = torch.float32 if args.unet_stream == "f32" else torch.float16
    if args.precision is not None:
        pipe.unet.precision = args.precision
    unet_f32 = pipe.unet.stream_f32()
    pipe.cfg_shared_input = not args.no_cfg_share
    pipe.shard_windows = args.shard_windows
    pipe.shard_cfg = args.shard_cfg
    if args.overlap_streams is not None:
        pipe.overlap_streams = args.overlap_streams
    n_overlap = pipe.overlap_streams
    pipe.overlap_split_cfg = args.overlap_split_cfg
    overlapped = n_overlap > 1 and not args.shard_windows and (args.frames > 8 or args.overlap_split_cfg)
    if args.shard_cfg and not args.shard_windows:
        raise SystemExit("--shard-cfg refines --shard-windows")
    clip = synthetic_clip(args.frames, args.height, args.width, seed=0 if args.shard_windows else rank, dev=dev)
    flows, psteps = None, []
    if args.propagation:
        from uav import init_weights
        from models_video.RAFT.raft_bi import RAFT_bi
        from models_video.propagation_module import Propagation
        raft = RAFT_bi(model_path=None, device=dev)
        init_weights.random_init_(raft, seed=777)
        with torch.no_grad():
            for name in ("conv2.weight", "conv2.bias"):     # damp the random flow head: few-pixel flows
                getattr(raft.fix_raft.update_block.flow_head.conv2, name.split(".")[1]).mul_(0.05)
        torch.cuda.synchronize(); t_r = time.perf_counter()
        raft_flows = list(raft.forward_slicing(clip, iters=20))
        torch.cuda.synchronize(); raft_s = time.perf_counter() - t_r
        pipe.propagator = Propagation(4, learnable=False)
        psteps = [s for s in (24, 26, 28) if s < args.ddim_steps]
        # A random-weight RAFT produces flows that fail the forward-backward check on every pixel (measured, round 4): the
        # propagation is then the identity and the line would time a no-op.  The flows that ENTER the pipeline are therefore exact
        # translations in forward_slicing's output contract ((1, T-1, 2, H, W), pixels per frame; backward = -forward): consistent
        # by construction, so the mask keeps everything but the border the warp leaves.  RAFT itself still runs (and is timed, outside
        # the timed region like inference_upscale_a_video.py:191); what the propagation changes is measured on a probe tensor.
        ff = torch.zeros_like(raft_flows[0]); ff[:, :, 0] = 2.0; ff[:, :, 1] = 1.0
        flows = [ff, -ff]
        with torch.no_grad():
            probe = torch.randn(1, 4, args.frames, args.height, args.width, device=dev)
            moved = pipe.propagator(probe, flows[0], flows[1], interpolation="nearest", mode="fuse", fuse_scale=0.5, alpha1=0.001, alpha2=0.05)
            prop_changed = float((moved != probe).float().mean().item())
            raft_moved = pipe.propagator(probe, raft_flows[0], raft_flows[1], interpolation="nearest", mode="fuse", fuse_scale=0.5, alpha1=0.001, alpha2=0.05)
            raft_changed = float((raft_moved != probe).float().mean().item())
        del probe, moved, raft_moved
        if prop_changed <= 0.5:
            raise SystemExit(f"bench.py --propagation: the flows change only {prop_changed:.3f} of the latent elements (need > 0.5)")
    kw = dict(image=clip, flows_bi=flows, num_inference_steps=args.ddim_steps, guidance_scale=6.0, noise_level=120,
              negative_prompt="blur, worst quality", propagation_steps=psteps)
    prompt = "best quality, extremely detailed"

    ncl = max(1, args.clips_per_step)
    if ncl > 1 and args.shard_windows:
        raise SystemExit("--clips-per-step > 1 and --shard-windows are exclusive")
    if ncl > 1:
        args.no_kernel_events = True
    _multi = {}

    def multi_setup(n):
        """n clips upscaled CONCURRENTLY on this GPU: one HIP stream + host thread + pipeline shell (own scheduler) each; UNet /
        VAE / text encoder are shared."""
        if n not in _multi:
            import copy
            streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
            pipes = []
            for _ in range(n):
                sh = copy.copy(pipe)
                sh.scheduler = copy.deepcopy(pipe.scheduler)
                pipes.append(sh)
            clips = [synthetic_clip(args.frames, args.height, args.width, seed=rank * n + j, dev=dev) for j in range(n)]
            _multi[n] = (streams, pipes, clips)
        return _multi[n]

    def new_generator(seed):
        # a DEVICE generator, like the reference CLI's `torch.Generator(device=UAV_device).manual_seed(10)` (inference_upscale_a_video.py
        # :197): the two noise tensors of a clip (5.7 M normals) are drawn on the GPU inside the timed region.  A host generator
        # (UAV_BENCH_CPU_GENERATOR=1, what rounds 1-3 timed) draws them in fp16 on the host cores and copies them: ~0.1-0.2 s per clip
        if os.environ.get("UAV_BENCH_CPU_GENERATOR") == "1":
            return torch.Generator().manual_seed(seed)
        return torch.Generator(device=dev).manual_seed(seed)

    def one_step(seed, n=None):
        n = ncl if n is None else n
        if n == 1:
            gen = new_generator(seed)
            return pipe(prompt, generator=gen, **kw).images
        import threading
        streams, pipes, clips = multi_setup(n)
        outs = [None] * n
        errs = []

        def work(j):
            try:
                with torch.cuda.stream(streams[j]):
                    gen = new_generator(seed * n + j)
                    outs[j] = pipes[j](prompt, generator=gen, **{**kw, "image": clips[j]}).images
            except Exception as e:       # noqa: BLE001 — re-raised on the main thread
                errs.append(e)
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        th = [threading.Thread(target=work, args=(j,)) for j in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        for st in streams:
            cur.wait_stream(st)
        assert all(bool(torch.isfinite(o).all()) for o in outs)
        return outs[0]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            if same_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = one_step(100 + i)
    barrier()
    use_events = (not args.no_kernel_events) and rank == 0
    all_events = args.all_kernel_events or bool(os.environ.get("UAV_BENCH_DETAIL"))
    # window-sharded steps hold collectives: an extra step on rank 0 alone would hang, so that mode keeps its events inside the timed region
    timed_events = use_events and (args.timed_kernel_events or args.all_kernel_events or args.shard_windows)
    if use_events:
        ops.PROFILER.detail = bool(os.environ.get("UAV_BENCH_DETAIL"))
        if timed_events and not overlapped:   # per-launch events mean nothing while launches of two streams share the chip
            ops.PROFILER.start(only=None if all_events else {"conv_gemm"})
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(10 + i)
    barrier()
    elapsed = time.perf_counter() - t0
    peak_alloc, peak_reserved = torch.cuda.max_memory_allocated(dev), torch.cuda.max_memory_reserved(dev)
    ops.PROFILER.stop()
    timed_summary = ops.PROFILER.summary() if timed_events else None
    extra_summary = None
    if use_events and (overlapped or not timed_events or (not all_events and world == 1)):
        # per-kernel table: one extra step AFTER the timed region with events around every kernel family (serial, i.e. with the
        # stream overlap switched off for this step: the kernel measurements are those of the default mode)
        pipe.overlap_streams = 0
        ops.PROFILER.start()
        one_step(10)
        torch.cuda.synchronize()
        ops.PROFILER.stop()
        extra_summary = ops.PROFILER.summary()
        pipe.overlap_streams = n_overlap
        if overlapped or not timed_events:
            timed_summary = extra_summary
    # Throughput mode BESIDE the serial headline: two clips per GPU on concurrent HIP streams — the HBM-bound
    # kernels of one clip (GroupNorm / LayerNorm passes, the epilogue-bound short-K linears) run beside the power-limited MFMA
    # kernels of the other.  Measured after the timed region (1 warm-up step + 2 timed steps of 2 clips); never the headline value.
    assert out.shape == (1, 3, args.frames, 4 * args.height, 4 * args.width) and bool(torch.isfinite(out).all())
    two_clip = None
    if world == 1 and ncl == 1 and not args.no_throughput_mode and not args.shard_windows:
        one_step(300, 2)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for i in range(2):
            one_step(310 + i, 2)
        torch.cuda.synchronize(); e2 = time.perf_counter() - t2
        two_clip = {"clips_per_step": 2, "steps": 2, "ms_per_step": e2 / 2 * 1e3, "frames_per_s": 2 * 2 * args.frames / e2,
                    "note": "two clips per GPU on concurrent HIP streams (serving mode, `--clips-per-step 2` times it as the main "
                            "line); reported beside the serial headline, not instead of it"}
    assert out.shape == (1, 3, args.frames, 4 * args.height, 4 * args.width) and bool(torch.isfinite(out).all())
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device="cpu" if same_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        frames_total = (1 if args.shard_windows else world) * args.steps * args.frames * ncl
        res = {
            "metric": METRIC, "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.shard_windows else "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"configs[{3 if args.shard_windows else 4 if args.video_vae else 2 if args.propagation else 1}]"
                                   + (" (one spatial tile of a 540p clip)" if args.video_vae else "") + f": {args.frames}-frame {args.height}x{args.width}->{4 * args.height}x{4 * args.width}, "
                                   f"{args.ddim_steps} DDIM steps, guidance 6, noise_level 120, "
                                   + ("UNet on an fp32 residual stream (fp16 MFMA operands, fp32 latents), " if unet_f32 else
                                      "UNet on fp16 rows (the reference's .half() arithmetic), ")
                                   + ("CLIP ViT-H text tower on HIP kernels (one encode per distinct prompt pair), " if args.text_encoder == "clip" else "stand-in text embedding, ")
                                   + ("vae_video (" if args.video_vae else "vae_3d (")
                                   + ("all-fp16 decoder rows" if args.vae_fp16 else "fp32 residual stream, fp16 MFMA operands") + "), "
                                   + (f"RAFT flows (20 iters, {raft_s * 1e3:.0f} ms, outside the timed region like the reference) + "
                                      f"latent propagation at steps {psteps}; " if args.propagation else "no propagation; ")
                                   + (("ONE clip, (temporal window x guidance branch) units" if args.shard_cfg else "ONE clip, temporal windows")
                                      + " + decode chunks dealt over the ranks, all-gather per DDIM step (RCCL)"
                                      if args.shard_windows else
                                      (f"the clip's {'guidance branches / ' if args.overlap_split_cfg else ''}windows and decode chunks on {n_overlap} concurrent HIP streams, " if overlapped else "")
                                      + f"{ncl} clip(s) per GPU per step"
                                      + (" on concurrent HIP streams" if ncl > 1 else "") + " (clip-parallel, no collective)"),
                       "clips_per_step": world * ncl, "frames_per_clip": args.frames},
        }
        res["config"]["predicted_scaling"] = predicted_scaling(args, world, res["value"])
        res["config"]["overlap_mode"] = getattr(pipe, "last_overlap_mode", "serial")
        if args.propagation:
            res["config"]["propagation_changed_fraction"] = prop_changed
            res["config"]["propagation_flows"] = ("exact translation (2, 1) px / frame in forward_slicing's contract; flows of the random-weight "
                                                   f"RAFT (computed and timed, {raft_s * 1e3:.0f} ms) change {raft_changed:.3f} of the elements")
        if args.precision is not None:
            res["config"]["unet_precision"] = args.precision
        # device memory of the timed region (torch caching allocator, rank 0): peak bytes in live tensors / held by the allocator's
        # pools — side streams (overlap_streams, clips_per_step) own pools of their own
        res["config"]["peak_memory_gb"] = {"allocated": round(peak_alloc / 2 ** 30, 2), "reserved": round(peak_reserved / 2 ** 30, 2)}
        if two_clip is not None:
            res["throughput_mode"] = two_clip
        if args.digest:
            import hashlib
            res["config"]["output_sha256"] = hashlib.sha256(out.detach().float().cpu().numpy().tobytes()).hexdigest()
        if use_events:
            summ = timed_summary
            if ops.PROFILER.detail:                       # per-shape table to stderr, then fold back
                tot = sum(v["seconds"] for v in summ.values())
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["seconds"])[:90]:
                    print(f"{v['seconds'] * 1e3:9.2f} ms {100 * v['seconds'] / tot:5.1f}% n={v['launches']:5d} "
                          f"{(v['flops'] / v['seconds'] / 1e12 if v['flops'] else 0):7.1f} TF/s "
                          f"{v['bytes'] / v['seconds'] / 1e9:8.1f} GB/s  {k}", file=sys.stderr)
                folded = {}
                for k, v in summ.items():
                    kk = k.split(" ")[0]
                    d = folded.setdefault(kk, dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0))
                    for f in d:
                        d[f] += v[f]
                summ = folded
            dom = max(summ.items(), key=lambda kv: kv[1]["seconds"])
            name, d = dom
            ach = d["flops"] / d["seconds"] / 1e12
            table, table_steps = (summ, args.steps) if extra_summary is None else (extra_summary, 1)
            total_s = sum(v["seconds"] for v in table.values())
            share = table[name]["seconds"] / total_s
            # HBM traffic per launch of the dominant kernel cannot be read from inside this process: it comes from the
            # committed PMC pass over this same command (tools/pmc_traffic.sh -> profiles/pmc_conv_traffic.json;
            # TCC_EA0_RDREQ / WRREQ with the gfx950 corrections of MI355X_MICROARCH.md) and stays null without it.
            traffic, traffic_note = None, None
            tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_conv_traffic.json")
            if name == "conv_gemm" and os.path.exists(tfile) and not (args.propagation or args.shard_windows or args.video_vae or args.frames != 8 or args.height != 320 or args.width != 320 or args.unet_stream == "f16"):
                from uav import build as _build
                tj = json.load(open(tfile))
                if tj.get("kernel_sources_digest") == _build.conv_kernel_digest():
                    traffic = tj.get("hbm_bytes_per_launch")
                else:           # the PMC pass describes another build of the kernels: do not replay it
                    traffic_note = "profiles/pmc_conv_traffic.json was taken with other kernel sources (digest mismatch): not replayed"
            res["roofline"] = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                               "frac": ach / PEAK_TFLOPS_F16, "traffic": traffic,
                               "traffic_source": traffic_note if traffic is None else "replayed from profiles/pmc_conv_traffic.json (separate rocprofv3 "
                                                 "--pmc pass over this command with the SAME kernel sources — digest checked —, tools/pmc_traffic.sh); "
                                                 "not measured by this run",
                               "measured": ("one extra SERIAL step after the timed region (launches of the timed region overlap on "
                                            f"{n_overlap} streams)" if overlapped else "HIP events inside the timed region" if timed_events else
                                            "HIP events around every launch of one extra clip issued right behind the timed region (same process, "
                                            "same box, same inputs): the timed region itself carries no events since round 6 — they cost 0.63 % "
                                            "(profiles/r06_event_overhead_ab_*.jsonl; --timed-kernel-events restores the old placement)"),
                               "launches": d["launches"],
                               "avg_launch_us": d["seconds"] / d["launches"] * 1e6,
                               "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                               "flops_counted": "2*M*N*taps*C_in of every launch as issued: the upsamplers run as four 2x2 "
                                                "sub-pixel phase convs and count 16 of the reference's 36 multiply-adds per "
                                                "output element (SURVEY's 5608.7 TFLOP/clip counts 36); temporal taps that fall "
                                                "outside the clip (zero padding, skipped by the kernel) are NOT counted",
                               "kernel_time_share": share}
            # per kernel: MFMA-bound ones against the dense fp16 peak, HBM-bound ones (algorithmic bytes: every operand
            # read / written once) against the 8 TB/s HBM3E peak
            hbm_bound = ("groupnorm_stats", "groupnorm_finalize_fused", "groupnorm_apply", "layernorm", "temporal_attention", "attention_d64",
                         "cast_f16", "cast_hilo")
            res["kernel_breakdown"] = {k: {"launches": v["launches"], "ms": round(v["seconds"] * 1e3, 2),
                                           "tflops": round(v["flops"] / v["seconds"] / 1e12, 1) if v["flops"] else None,
                                           "GBps": round(v["bytes"] / v["seconds"] / 1e9, 1),
                                           "bound": "hbm" if k in hbm_bound else "mfma",
                                           "frac": round(v["bytes"] / v["seconds"] / 1e9 / PEAK_HBM_GBPS, 3) if k in hbm_bound
                                           else (round(v["flops"] / v["seconds"] / 1e12 / PEAK_TFLOPS_F16, 3) if v["flops"] else None)}
                                       for k, v in sorted(table.items(), key=lambda kv: -kv[1]["seconds"])}
            res["kernel_breakdown_source"] = ("HIP events around every kernel inside the timed region" if extra_summary is None else
                                              "one extra fully instrumented step after the timed region (the timed region itself carries "
                                              + ("events around conv_gemm only" if timed_events else "no events") + "); launches / ms are per that one step")
            res["kernel_time_ms_per_step"] = total_s / table_steps * 1e3
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
