"""Round-4 parity evidence on the GPU (VERDICT r3 "next" #1): BASELINE configs[1] and configs[2] END TO END at the shape
bench.py times — 8 frames 320x320 -> 1280x1280, 30 DDIM steps, guidance 6, full width (691 M-parameter UNet, vae_3d) —
against the fp32 oracle pipeline executed on the same GPU (oracle/gpu_shim.py: ATen fp32 kernels, convolutions as
exact-fp32 GEMMs; test infrastructure, never the product path).

  * configs[1]: per-step latents curve over the whole schedule, `.images` as all-pixel AND unsaturated rel-L2;
  * configs[2]: the same with flow-guided propagation at DDIM steps 24/26/28 on consistent synthetic flows fed to BOTH
    sides (RAFT itself is pinned in tests/test_raft_gpu.py; see build_flows), the oracle replays steps 0..23 from the
    configs[1] run (identical until the first propagation step) — reference pipeline_upscale_a_video.py:651-657,
    propagation_module.py:104-135,194-281 in fp32, which is the run an fp32-latent engine is held to.

Common noise: the two randn draws (pipeline:547,567) come from one CPU generator in fp32 on both sides.
Cost: ~30 x 3.5 s + 6 x 3.5 s + 6 x 1.7 s of oracle time, ~20 s of engine time.  Every number goes to
gpurun_out/parity.jsonl (copied to profiles/r04_parity_*.jsonl).
"""
import json
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

T, H, W = 8, 320, 320
STEPS, GUIDANCE, NOISE_LEVEL = 30, 6.0, 120
PROMPT, NEGATIVE = "best quality, extremely detailed", "blur, worst quality"
PROP_STEPS = [24, 26, 28]


def rel_l2(a, b):
    a = a.float(); b = b.float().to(a.device)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def report(name, **vals):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(case=name, **vals)) + "\n")


def image_errors(out, ref):
    """(all-pixel rel-L2, rel-L2 over the pixels the reference does not clamp, clamped fraction)."""
    unsat = ref.abs() < 0.999
    return rel_l2(out, ref), rel_l2(out[unsat], ref[unsat]), 1.0 - unsat.float().mean().item()


def build_models(dev):
    import synth
    from uav import configs
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.unet_video import UNetVideoModel
    unet = UNetVideoModel.from_config(dict(configs.UNET_VIDEO))
    usd = synth.synth_state_dict(unet.state_dict(), seed=1234)
    unet.load_state_dict(usd, strict=True)
    unet = unet.half().to(dev).eval()
    vae = AutoencoderKLVideo.from_config(dict(configs.VAE_3D))
    vsd = synth.synth_state_dict(vae.state_dict(), seed=4321)
    vae.load_state_dict(vsd, strict=True)
    vae = vae.to(dev).eval()
    return unet, usd, vae, vsd


def build_flows(dev, t=T, h=H, w=W):
    """Consistent bidirectional flows at the latent resolution (golden_cases.consistent_flows: the forward-backward check
    passes, frames really are warped, one band is masked out).  Round 4, call 1 measured that flows of a random-weight RAFT
    (what bench.py --propagation feeds) fail the consistency check everywhere — the propagation is then the identity and tests
    nothing; RAFT itself is pinned in tests/test_raft_gpu.py against the reference's fixtures."""
    import golden_cases as GC
    ff, fb = GC.consistent_flows(t, h, w)
    return [ff.to(dev).contiguous(), fb.to(dev).contiguous()]


def engine_run(dev, unet, vae, clip, flows=None, prop_steps=(), steps=STEPS):
    import golden_cases as GC
    from uav import configs
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.propagation_module import Propagation
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    tok = StandInTokenizer()
    dim = configs.UNET_VIDEO["cross_attention_dim"]
    prop = Propagation(4, learnable=False) if flows is not None else None
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, dim, dtype=torch.float32), tokenizer=tok,
                                low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED), vae=vae, unet=unet,
                                propagator=prop).to(dev)
    pipe.latents_trace = []
    torch.cuda.synchronize(); t0 = time.time()
    out, lat = pipe(PROMPT, image=clip.to(dev), flows_bi=flows, generator=torch.Generator().manual_seed(10),
                    num_inference_steps=steps, guidance_scale=GUIDANCE, noise_level=NOISE_LEVEL, negative_prompt=NEGATIVE,
                    propagation_steps=list(prop_steps), return_dict=False)
    torch.cuda.synchronize()
    return dict(images=out, latents=lat, trace=pipe.latents_trace, seconds=time.time() - t0)


def oracle_inputs(dev, clip):
    import synth
    from uav import configs
    dim = configs.UNET_VIDEO["cross_attention_dim"]
    gen = torch.Generator().manual_seed(10)
    lr_noise = torch.randn(clip.shape, generator=gen)
    lat0 = torch.randn((1, 4, T, H, W), generator=gen)
    pe = torch.cat([synth.synth_prompt_embeds(NEGATIVE, dim), synth.synth_prompt_embeds(PROMPT, dim)])
    return lr_noise.to(dev), lat0.to(dev), pe.to(dev)


def oracle_runs(dev, usd, vsd, clip, flows):
    """configs[1] over the whole schedule, then configs[2] resumed at the first propagation step."""
    import golden_cases as GC
    import gpu_shim
    from uav import configs
    lr_noise, lat0, pe = oracle_inputs(dev, clip)
    usd_d = {k: v.to(dev) for k, v in usd.items()}
    vsd_d = {k: v.to(dev) for k, v in vsd.items()}
    kw = dict(num_inference_steps=STEPS, guidance_scale=GUIDANCE, noise_level=NOISE_LEVEL, lr_noise=lr_noise, latents=lat0,
              scheduler_kwargs=GC.SCHED, return_trace=True)
    with gpu_shim.oracle_on(dev) as O:
        torch.cuda.synchronize(); t0 = time.time()
        img1, lat1, tr1 = O.pipeline_call(usd_d, configs.UNET_VIDEO, vsd_d, configs.VAE_3D, clip.to(dev), pe, **kw)
        torch.cuda.synchronize(); s1 = time.time() - t0
        i0 = PROP_STEPS[0]
        img2, lat2, tr2 = O.pipeline_call(usd_d, configs.UNET_VIDEO, vsd_d, configs.VAE_3D, clip.to(dev), pe,
                                          flows_bi=flows, propagation_steps=tuple(PROP_STEPS), resume=(i0, tr1[i0 - 1]), **kw)
        torch.cuda.synchronize(); s2 = time.time() - t0 - s1
    return dict(c1=dict(images=img1, latents=lat1, trace=tr1, seconds=s1),
                c2=dict(images=img2, latents=lat2, trace=tr1[:i0] + tr2, seconds=s2))


@pytest.fixture(scope="module")
def headline(dev):
    import synth
    clip = synth.synth_clip(1, T, H, W, seed=3, motion=(2, 1))
    unet, usd, vae, vsd = build_models(dev)
    flows = build_flows(dev)
    eng1 = engine_run(dev, unet, vae, clip)
    eng2 = engine_run(dev, unet, vae, clip, flows=flows, prop_steps=PROP_STEPS)
    torch.cuda.empty_cache()
    ora = oracle_runs(dev, usd, vsd, clip, flows)
    save = os.environ.get("UAV_R4_SAVE_ORACLE")          # tools/r4/parity_variants.py re-uses the oracle side in the same box
    if save:
        torch.save(dict(clip=clip, flows=[f.cpu() for f in flows],
                        c1=dict(images=ora["c1"]["images"].cpu(), latents=ora["c1"]["latents"].cpu()),
                        c2=dict(images=ora["c2"]["images"].cpu(), latents=ora["c2"]["latents"].cpu())), save)
    return dict(eng1=eng1, eng2=eng2, ora=ora, flows=flows)


def _check(name, eng, ora, bars):
    curve = [rel_l2(e, o) for e, o in zip(eng["trace"], ora["trace"])]
    assert len(curve) == STEPS
    e_lat = rel_l2(eng["latents"], ora["latents"])
    e_all, e_unsat, sat = image_errors(eng["images"], ora["images"])
    report(name, latents_rel_l2_per_step=curve, latents_rel_l2=e_lat, images_rel_l2_all_pixels=e_all,
           images_rel_l2_unsaturated=e_unsat, images_saturated_fraction=sat, engine_seconds=eng["seconds"],
           oracle_seconds=ora["seconds"], shape=[T, H, W], steps=STEPS, guidance=GUIDANCE)
    assert eng["images"].shape == ora["images"].shape == (1, 3, T, 4 * H, 4 * W)
    assert bool(torch.isfinite(ora["images"]).all()) and bool(torch.isfinite(eng["images"]).all())
    assert e_lat < bars[0], (name, e_lat, curve)
    assert e_all < bars[1], (name, e_all, e_unsat)
    return curve, e_lat, e_all, e_unsat


def test_headline_configs1_end_to_end_vs_gpu_oracle(headline):
    """BASELINE configs[1] (pipeline_upscale_a_video.py:436-716): latents after 30 steps inside the stated 1e-3."""
    # measured (round 4, call 1): latents 9.0e-4, images 1.075e-3 (all pixels) / 1.39e-3 (unsaturated) without the samplers'
    # hi|lo operands; with them (default since) 8.2e-4, 9.5e-4 / 1.23e-3.  Both bars are BASELINE.json's stated 1e-3.
    # round 6 (block tails as hi|lo pairs by default, fused cross-attention sub-layers): 7.3e-4, 8.7e-4 / 1.12e-3 — bars 8.0e-4 / 9.2e-4
    # (the stated 1e-3 with a margin instead of 5 % under it)
    _check("r4_headline_configs1_8x320x320_30steps", headline["eng1"], headline["ora"]["c1"], bars=(8.0e-4, 9.2e-4))


def test_headline_configs2_propagation_end_to_end_vs_gpu_oracle(headline):
    """BASELINE configs[2]: + RAFT flows and fp32 flow-guided propagation at steps 24/26/28 (same flows on both sides)."""
    eng, ora = headline["eng2"], headline["ora"]["c2"]
    curve, e_lat, e_all, e_unsat = _check("r4_headline_configs2_8x320x320_30steps_propagation", eng, ora, bars=(8.0e-4, 9.2e-4))    # measured 6.7e-4 / 8.7e-4
    # the propagation did something (vs the no-propagation run), and a nearest-neighbour index flip would show as an O(1)
    # difference on single elements: count them
    moved = rel_l2(eng["latents"], headline["eng1"]["latents"])
    off = ((eng["latents"].float() - ora["latents"].float().to(eng["latents"].device)).abs() > 5e-2).float().mean().item()
    report("r4_headline_configs2_propagation_effect", latents_rel_l2_vs_no_propagation=moved, fraction_of_latent_elements_off_by_5e_2=off,
           flow_absmax=float(max(f.abs().max().item() for f in headline["flows"])))
    assert moved > 5e-3, moved
    assert off < 1e-3, off


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["propagation_nearest_f32_ties", "propagation_nearest_f32_wide", "propagation_bilinear_f32_wide"])
def test_propagation_fp32_latents_vs_reference_fp32_run(dev, name):
    """fp32 latents (the default, fp32-stream pipeline) are warped as fp32 values on fp32 grids
    (`uav_propagate_step_f32`) and held to the reference's fp32 run on inputs that sit ON the .5 rounding ties of the
    nearest-neighbour warp (tests/golden/propagation_*_f32_*.pt, reference `Propagation` on CPU fp32 tensors).  The fp16
    replay (round 3's behaviour on fp32 latents: values and coordinates rounded to fp16) picks another source pixel on
    ~half of the tie pixels — the test discriminates the two."""
    import golden_cases as GC
    from models_video.propagation_module import Propagation
    kind, t, h, w, interp = GC.PROP_HALF_CASES[name.replace("_f32", "_half")]
    x, ff, fb = GC.prop_half_inputs(kind, t, h, w)
    gold = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"))
    prop = Propagation(4, learnable=False)
    kw = dict(interpolation=interp, mode="fuse", fuse_scale=0.5, alpha1=0.001, alpha2=0.05)
    out = prop(x.to(dev), ff.to(dev), fb.to(dev), **kw)
    assert out.dtype == torch.float32 and out.shape == gold.shape
    diff = (out.cpu() - gold).abs()
    off = (diff > 1e-5).float().mean().item()
    identical = (out.cpu() == gold).float().mean().item()
    prop.coord_f16 = True                                   # round 3's hybrid on the same fp32 latents
    out16 = prop(x.to(dev), ff.to(dev), fb.to(dev), **kw)
    off16 = ((out16.float().cpu() - gold).abs() > 1e-2).float().mean().item()
    report("r4_" + name, fraction_off_by_1e_5_vs_reference_fp32=off, bit_identical_fraction=identical,
           fp16_replay_fraction_off_by_1e_2_vs_reference_fp32=off16, max_abs=diff.max().item())
    assert off < 2e-4, (off, identical)                     # bilinear: fma / summation-order ulps only; nearest: bit-identical
    if interp == "nearest":
        assert identical > 0.9998, identical
        assert off16 > 0.3, off16                           # the fp16 replay is a different function on tie pixels


# ------------------------------------------------------------------------------------------------
def test_configs2_full_width_30_steps_vs_reference_pipeline(dev):
    """BASELINE configs[2] against the REFERENCE'S OWN pipeline (not the oracle): full width, 8 frames 64x64 -> 256x256, 30 DDIM
    steps, guidance 6, the reference `Propagation` at loop indices 24 / 26 / 28 on consistent flows, everything fp32
    (tests/golden/pipe_full30_64_prop.pt, `oracle/make_golden.py --full30prop`; pipeline_upscale_a_video.py:651-657,
    propagation_module.py:194-281).  Common noise: fp32 draws from one CPU generator."""
    import golden_cases as GC
    import synth
    path = os.path.join(ROOT, "tests", "golden", "pipe_full30_64_prop.pt")
    if not os.path.exists(path):
        pytest.skip("fixture pipe_full30_64_prop.pt not generated")
    gold = torch.load(path)
    pc = GC.FULL_CASES["pipe_full30_64_prop"]
    assert (pc["prompt"], pc["negative"], pc["guidance"], pc["noise_level"]) == (PROMPT, NEGATIVE, GUIDANCE, NOISE_LEVEL)
    unet, usd, vae, vsd = build_models(dev)
    clip = synth.synth_clip(1, pc["t"], pc["h"], pc["w"], seed=pc["clip_seed"])
    flows = build_flows(dev, pc["t"], pc["h"], pc["w"])
    eng = engine_run(dev, unet, vae, clip, flows=flows, prop_steps=pc["propagation_steps"], steps=pc["steps"])
    steps = list(gold["steps"])
    curve = [rel_l2(eng["trace"][k - 1], gold["latents_fp32"][i]) for i, k in enumerate(steps)]
    g = gold["images_fp32_sub2"].float()
    img = eng["images"].float().cpu()[..., ::2, ::2]
    e_all, e_unsat, sat = image_errors(img, g)
    report("r4_pipe_full30_64_prop_vs_reference_pipeline", steps=steps, latents_rel_l2_at_kept_steps=curve,
           images_rel_l2_all_pixels=e_all, images_rel_l2_unsaturated=e_unsat, images_saturated_fraction=sat)
    assert curve[-1] < 1.0e-3, curve                    # latents after the whole schedule incl. the three propagation sweeps
    assert e_all < 1.3e-3, (e_all, e_unsat)             # 64x64 proxy (the headline-shape bars are the stated 1e-3, above)
