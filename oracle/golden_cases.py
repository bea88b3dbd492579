"""TEST INFRASTRUCTURE ONLY.  Reduced-width configs and seeded inputs shared by
oracle/make_golden.py (reference side, build container) and tests/ (engine side, GPU box).

The configs keep the reference's full topology (4 down / mid / 4 up blocks, temporal modules
everywhere, cross-only + self attention, 32 GroupNorm groups) at 1/4 width, so every code path
of the north-star model is exercised while the CPU reference runs in seconds.
`attention_head_dim` is the head COUNT in this model: 2 heads -> head dims 64 /
64 / 128, the head dims of the full model.
"""
import torch

import synth

UNET_TINY = {
    "act_fn": "silu", "attention_head_dim": 2, "block_out_channels": [64, 128, 128, 256],
    "center_input_sample": False, "cross_attention_dim": 64,
    "down_block_types": ["DownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D"],
    "downsample_padding": 1, "dual_cross_attention": False, "flip_sin_to_cos": True, "freq_shift": 0,
    "in_channels": 7, "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05,
    "norm_num_groups": 32, "num_class_embeds": 1000, "only_cross_attention": [True, True, True, False],
    "out_channels": 4, "sample_size": 128,
    "up_block_types": ["CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "UpBlock3D"],
    "use_linear_projection": True, "down_temporal_idx": [0, 1, 2, 3], "mid_temporal": True,
    "up_temporal_idx": [0, 1, 2, 3], "temporal_module_config": {"attention_block_types": ["", ""]},
}

VAE3D_TINY = {
    "act_fn": "silu", "block_out_channels": [64, 64, 128],
    "down_block_types": ["DownEncoderBlock3D"] * 3, "in_channels": 3, "latent_channels": 4,
    "layers_per_block": 2, "norm_num_groups": 32, "out_channels": 3, "sample_size": 256,
    "up_block_types": ["UpDecoderBlock3D"] * 3, "scaling_factor": 0.08333,
}

VAEVIDEO_TINY = dict(VAE3D_TINY, up_block_types=["UpDecoderBlock3D_plus"] * 3, condition_img=True,
                     condition_channels=64, use_temporal_block=True)

# upstream SD-x4-upscaler scheduler_config.json values (absent from the reference tree)
SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
             clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="v_prediction")


def _g(seed):
    return torch.Generator().manual_seed(seed)


def unet_inputs(bsz, t, h, w, cross_dim, seed=100):
    g = _g(seed + t * 7 + h)
    sample = torch.randn(1, 4, t, h, w, generator=g).half().float().repeat(bsz, 1, 1, 1, 1)
    low = torch.randn(1, 3, t, h, w, generator=g).half().float().repeat(bsz, 1, 1, 1, 1)
    ehs = torch.randn(bsz, 77, cross_dim, generator=g).half().float()
    return sample, low, ehs, 925, torch.tensor([120], dtype=torch.long)


def vae_inputs(bsz, t, h, w, seed=200):
    g = _g(seed + t + h)
    z = (torch.randn(bsz, 4, t, h, w, generator=g) * 2.0).half().float()
    img = synth.synth_clip(bsz, t, h, w, seed=seed)
    return z, img


def prop_inputs(t, h, w, seed=300):
    """x0 latents + bidirectional flows of a (2,1) px/frame translation with noise and an
    occluded band; sub-pixel parts of .3 keep nearest-neighbour rounding away from .5 ties."""
    g = _g(seed)
    x = torch.randn(1, 4, t, h, w, generator=g).half().float()
    ff = torch.zeros(1, 2, t - 1, h, w); fb = torch.zeros(1, 2, t - 1, h, w)
    ff[:, 0] = 2.3; ff[:, 1] = 1.3; fb[:, 0] = -2.3; fb[:, 1] = -1.3
    ff = ff + 0.02 * torch.randn(ff.shape, generator=g); fb = fb + 0.02 * torch.randn(fb.shape, generator=g)
    fb[:, :, :, h // 3: h // 2] += 3.0                      # inconsistent region -> mask = 0
    return x, ff.half().float(), fb.half().float()


def prop_inputs_ties(t, h, w, seed=310):
    """All-half propagation inputs that sit ON the discontinuities of the nearest-neighbour warp: flow components with an
    exact .5 sub-pixel part (half of the pixels, the rest carry small noise), so x + flow lands on a rounding tie before
    the fp16 normalise / un-normalise round trip of flow_warp (propagation_module.py:123-132) moves it; wide frames
    (w = 320) add the fp16 quantisation of large coordinates (spacing .25 above 256)."""
    g = _g(seed + h + w)
    x = torch.randn(1, 4, t, h, w, generator=g).half().float()
    ff = torch.zeros(1, 2, t - 1, h, w); fb = torch.zeros(1, 2, t - 1, h, w)
    ff[:, 0] = 2.5; ff[:, 1] = 1.5; fb[:, 0] = -2.5; fb[:, 1] = -1.5
    noisy = (torch.rand(ff.shape, generator=g) < 0.5).float()
    ff = ff + noisy * 0.03 * torch.randn(ff.shape, generator=g); fb = fb + noisy * 0.03 * torch.randn(fb.shape, generator=g)
    fb[:, :, :, h // 3: h // 2] += 3.0
    return x, ff.half().float(), fb.half().float()


def consistent_flows(t, h, w, seed=4242):
    """Bidirectional flows (1,2,t-1,h,w) x 2 that PASS the forward-backward check (frames really are warped): a (2.3, 1.3)
    px/frame translation with small noise, fp16-representable, plus a band where the backward flow contradicts the forward one
    (mask = 0 there).  Flows of a random-weight RAFT fail the check everywhere (round 4, GPU call 1): propagation = identity."""
    g = _g(seed)
    ff = torch.zeros(1, 2, t - 1, h, w); fb = torch.zeros(1, 2, t - 1, h, w)
    ff[:, 0] = 2.3; ff[:, 1] = 1.3; fb[:, 0] = -2.3; fb[:, 1] = -1.3
    ff = ff + 0.02 * torch.randn(ff.shape, generator=g); fb = fb + 0.02 * torch.randn(fb.shape, generator=g)
    fb[:, :, :, h // 3: h // 2] += 3.0
    return ff.half().float(), fb.half().float()


# reference Propagation run on CPU *half* tensors (the dtype the pipeline hands it, pipeline:651): name -> (inputs fn, t, h, w, interp)
PROP_HALF_CASES = {
    "propagation_nearest_half": ("plain", 8, 24, 32, "nearest"),
    "propagation_bilinear_half": ("plain", 8, 24, 32, "bilinear"),
    "propagation_nearest_half_ties": ("ties", 5, 24, 40, "nearest"),
    "propagation_nearest_half_wide": ("ties", 4, 16, 320, "nearest"),
    "propagation_bilinear_half_wide": ("ties", 3, 16, 320, "bilinear"),
}


def prop_half_inputs(kind, t, h, w):
    return prop_inputs(t, h, w) if kind == "plain" else prop_inputs_ties(t, h, w)


PIPE_CASES = {
    # plumbing case of BASELINE config 1 shape family, scaled to run on CPU in seconds
    "pipe_t8_vae3d": dict(vae="vae3d", t=8, h=16, w=16, steps=3, guidance=6.0, noise_level=120,
                          prompt="best quality, extremely detailed", negative="blur, worst quality",
                          propagation_steps=()),
    # sliding-window branch (T > 8: windows [0,8) and [2,10)) + propagation + video VAE
    "pipe_t10_vaevideo_prop": dict(vae="vaevideo", t=10, h=16, w=24, steps=3, guidance=6.0, noise_level=120,
                                   prompt="best quality, extremely detailed", negative="blur, worst quality",
                                   propagation_steps=(1,)),
}


def pipeline_inputs(case, seed=400):
    image = synth.synth_clip(1, case["t"], case["h"], case["w"], seed=seed)
    flows = None
    if case["propagation_steps"]:
        _, ff, fb = prop_inputs(case["t"], case["h"], case["w"], seed=seed + 1)
        flows = [ff, fb]
    return image, flows


def colorfix_inputs(seed=500):
    """Decoded frames (content) and the bicubic-upsampled LR frames (style) of the CLI's colour fix: (T,3,H,W) fp32."""
    g = _g(seed)
    t, h, w = 2, 24, 20
    lr = synth.synth_clip(1, t, h, w, seed=seed)[0].permute(1, 0, 2, 3).contiguous()          # (T,3,h,w) in [-1,1]
    content = torch.nn.functional.interpolate(lr, scale_factor=4, mode="nearest") * 0.8 + 0.15 + \
        0.2 * torch.randn(t, 3, 4 * h, 4 * w, generator=g)
    return lr, content.clamp(-1, 1).contiguous()


# FULL-WIDTH cases (released architecture), generated by `make_golden.py --full` from the reference's own modules
FULL_CASES = {
    "unet_full_t8_64": (2, 8, 64, 64),
    "vae3d_full_t3_48": (1, 3, 48, 48),
    "vaevideo_full_t3_48": (1, 3, 48, 48),
    # BASELINE.json configs[0]: single 8-frame 128x128 -> 512x512 clip, 5 DDIM steps, no propagation
    "pipe_c1_full": dict(t=8, h=128, w=128, steps=5, guidance=6.0, noise_level=120, clip_seed=41,
                         prompt="best quality, extremely detailed", negative="blur, worst quality"),
    # the FULL 30-step schedule of BASELINE configs[1] at the released width, 64x64 so the CPU reference finishes in minutes
    "pipe_full30_64": dict(t=8, h=64, w=64, steps=30, guidance=6.0, noise_level=120, clip_seed=43,
                           prompt="best quality, extremely detailed", negative="blur, worst quality"),
    # BASELINE configs[2] at the released width: the same run with flow-guided propagation at DDIM steps 24, 26, 28 (0-based
    # loop indices, inference_upscale_a_video.py `-p 24,26,28`) on consistent_flows(8, 64, 64)
    "pipe_full30_64_prop": dict(t=8, h=64, w=64, steps=30, guidance=6.0, noise_level=120, clip_seed=43,
                                prompt="best quality, extremely detailed", negative="blur, worst quality",
                                propagation_steps=(24, 26, 28)),
    # BASELINE configs[3]'s schedule at the released width: T = 14 -> windows [0,8), [6,14) and the re-anchored duplicate
    # [6,14) of the reference loop (pipeline_upscale_a_video.py:601-635), epsilon blend over the shared frames, 30 steps
    "pipe_full30_64_t14": dict(t=14, h=64, w=64, steps=30, guidance=6.0, noise_level=120, clip_seed=47,
                               prompt="best quality, extremely detailed", negative="blur, worst quality"),
    # BASELINE configs[4]'s path at the released width: the reference CLI's tile loop (inference_upscale_a_video.py:207-304,
    # tile_size 64 -> two overlapping tiles 128 and 160 wide, H = 68 is not a multiple of 8) around the pipeline with the
    # full-width `vae_video` decoder (LR-frame conditioning, SFT fuse; vae_video.py:365-405), 5 steps, one shared generator
    "pipe_tiled_full_videovae": dict(t=3, h=68, w=160, tile=64, steps=5, guidance=6.0, noise_level=120, clip_seed=53,
                                     prompt="best quality, extremely detailed", negative="blur, worst quality"),
    # round 6 (VERDICT r5 missing #1): the same tile loop + `vae_video` decoder at the 30-step schedule BASELINE configs[4] runs
    # (the 5-step case above stays as a labelled stress case: 200-timestep strides weigh every UNet error ~6x)
    "pipe_tiled_full_videovae_30": dict(t=3, h=68, w=160, tile=64, steps=30, guidance=6.0, noise_level=120, clip_seed=53,
                                        prompt="best quality, extremely detailed", negative="blur, worst quality"),
}
