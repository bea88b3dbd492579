"""Round-3 parity evidence on the GPU (VERDICT r2 "next" #1):

  * the UNet's two residual-stream precisions (UNetVideoModel.stream_dtype: fp16 rows like the reference's `.half()` UNet, or
    fp32 rows with fp16 MFMA operands) against the reference's own full-width outputs;
  * BASELINE configs[1]'s REAL shapes — one UNetVideoModel.forward on (2,4,8,320,320) and one vae_3d decode chunk at
    320x320 -> 1280x1280 — against the fp32 oracle executed ON THE GPU (oracle/gpu_shim.py: ATen fp32 kernels, convolutions
    as exact-fp32 GEMMs; test-only), i.e. an independent checker at the shape bench.py times;
  * the whole 30-step schedule at the released width against the reference pipeline's per-step latents, in fp32 and in the
    CLI's half mix (tests/golden/pipe_full30_64.pt, `oracle/make_golden.py --full30`).

Bars are the measured values + 25 % (DESIGN.md §4 has the table); every number goes to gpurun_out/parity.jsonl.
"""
import json
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.float(); b = b.float().to(a.device)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def report(name, **vals):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(case=name, **vals)) + "\n")


def pinning(name):
    return json.load(open(os.path.join(GOLD, "PINNING.json")))["cases"][name]


STREAMS = {"fp16_stream": torch.float16, "fp32_stream": torch.float32}


@pytest.fixture(scope="module")
def full(dev):
    import golden_cases as GC
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
    return dict(GC=GC, synth=synth, configs=configs, unet=unet, vae=vae, usd=usd, vsd=vsd)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", list(STREAMS))
def test_unet_full_width_stream_modes_vs_reference(full, dev, mode):
    """691 M-parameter UNet, B2 x T8 x 64x64, against the reference module's fp32 output: fp16 rows 1.72e-3 (the
    reference's own `.half()` run: 1.88e-3), fp32 rows 1.01e-3 (tools/stream_numerics.py predicted both on CPU)."""
    GC = full["GC"]
    bsz, t, h, w = GC.FULL_CASES["unet_full_t8_64"]
    sample, low, ehs, ts, cl = GC.unet_inputs(bsz, t, h, w, full["configs"].UNET_VIDEO["cross_attention_dim"])
    unet = full["unet"]
    unet.stream_dtype = STREAMS[mode]
    try:
        with torch.no_grad():
            out = unet(sample.half().to(dev), ts, low.half().to(dev), encoder_hidden_states=ehs.half().to(dev), class_labels=cl).sample
            out32 = unet(sample.to(dev), ts, low.to(dev), encoder_hidden_states=ehs.half().to(dev), class_labels=cl).sample
    finally:
        unet.stream_dtype = None
    gold = torch.load(os.path.join(GOLD, "unet_full_t8_64.pt"))
    e32, e16, e32o = rel_l2(out, gold["fp32"]), rel_l2(out, gold["fp16"]), rel_l2(out32, gold["fp32"])
    report("r3_unet_full_t8_64_" + mode, rel_l2_vs_reference_fp32=e32, rel_l2_vs_reference_fp16_run=e16,
           rel_l2_vs_reference_fp32_with_fp32_output=e32o, reference_fp16_vs_fp32=pinning("unet_full_t8_64")["reference_fp16_vs_fp32_rel_l2"])
    assert out.dtype == torch.float16 and out32.dtype == torch.float32
    assert e32 < (1e-3 if mode == "fp32_stream" else 2.15e-3), e32      # measured 8.07e-4 / 1.707e-3
    assert e32o <= e32 + 1e-5


# ------------------------------------------------------------------------------------------------
def test_headline_shape_unet_forward_vs_gpu_oracle(full, dev):
    """BASELINE configs[1]: ONE UNetVideoModel.forward at the shape bench.py times — sample (2,4,8,320,320), low_res
    (2,3,8,320,320), 77 text tokens — against the fp32 oracle run on the same GPU (177 TFLOP of fp32 GEMMs)."""
    import gpu_shim
    GC = full["GC"]
    cfg = full["configs"].UNET_VIDEO
    sample, low, ehs, ts, cl = GC.unet_inputs(2, 8, 320, 320, cfg["cross_attention_dim"])
    unet = full["unet"]
    outs = {}
    for mode, dt in STREAMS.items():
        unet.stream_dtype = dt
        try:
            with torch.no_grad():
                torch.cuda.synchronize(); t0 = time.time()
                outs[mode] = unet(sample.to(dev), ts, low.to(dev), encoder_hidden_states=ehs.half().to(dev), class_labels=cl).sample
                torch.cuda.synchronize(); outs[mode + "_s"] = time.time() - t0
        finally:
            unet.stream_dtype = None
    torch.cuda.empty_cache()
    usd = {k: v.to(dev) for k, v in full["usd"].items()}
    with gpu_shim.oracle_on(dev) as O:
        torch.cuda.synchronize(); t0 = time.time()
        ref = O.unet_forward(usd, cfg, sample.to(dev), ts, low.to(dev), ehs.to(dev), cl.to(dev))
        torch.cuda.synchronize(); t_ora = time.time() - t0
    del usd
    torch.cuda.empty_cache()
    e16, e32 = rel_l2(outs["fp16_stream"], ref), rel_l2(outs["fp32_stream"], ref)
    report("r3_headline_unet_2x8x320x320_vs_gpu_oracle", fp16_stream_rel_l2=e16, fp32_stream_rel_l2=e32, oracle_seconds=t_ora,
           engine_seconds_first_call_fp16=outs["fp16_stream_s"], engine_seconds_first_call_fp32=outs["fp32_stream_s"],
           ref_absmean=ref.abs().mean().item())
    assert ref.shape == (2, 4, 8, 320, 320) and bool(torch.isfinite(ref).all())
    assert e32 < 9.6e-4, (e16, e32)                     # measured 7.68e-4: inside BASELINE.json's 1e-3 at the shape the bench times
    assert e16 < 2.1e-3, (e16, e32)                     # measured 1.68e-3
    assert e32 < e16


def test_headline_shape_vae_chunk_vs_gpu_oracle(full, dev):
    """BASELINE configs[1]: one vae_3d decode chunk (1,4,3,320,320) -> (1,3,3,1280,1280), incl. the d = 512 mid-block
    attention over L = 102 400 positions, against the fp32 oracle on the GPU (row-chunked softmax, exact)."""
    import gpu_shim
    GC = full["GC"]
    z, img = GC.vae_inputs(1, 3, 320, 320)
    vae = full["vae"]
    with torch.no_grad():
        out = vae.decode(z.to(dev), img.to(dev), 1.0).sample
    torch.cuda.empty_cache()
    vsd = {k: v.to(dev) for k, v in full["vsd"].items()}
    with gpu_shim.oracle_on(dev) as O:
        torch.cuda.synchronize(); t0 = time.time()
        ref = O.vae_decode(vsd, full["configs"].VAE_3D, z.to(dev), img.to(dev), 1.0)
        torch.cuda.synchronize(); t_ora = time.time() - t0
    e = rel_l2(out, ref)
    unsat = ref.abs() < 0.999
    e_un = rel_l2(out[unsat], ref[unsat])
    report("r3_headline_vae3d_chunk_3x320x320_vs_gpu_oracle", rel_l2=e, rel_l2_unsaturated=e_un, oracle_seconds=t_ora,
           saturated_fraction=1.0 - unsat.float().mean().item(), ref_absmean=ref.abs().mean().item())
    assert out.shape == ref.shape == (1, 3, 3, 1280, 1280)
    assert e < 7.3e-4, e                                # measured 5.8e-4


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", list(STREAMS))
def test_full_width_30_step_curve_vs_reference(full, dev, mode):
    """The whole 30-step schedule (8 frames 64x64 -> 256x256, guidance 6) at the released width against the reference
    pipeline's per-step latents: its fp32 run and its run in the CLI's half mix.  The reference's own half-vs-fp32 distance
    (PINNING.json) is the yardstick for the fp16-stream mode; the fp32-stream mode is measured against the fp32 run."""
    path = os.path.join(GOLD, "pipe_full30_64.pt")
    if not os.path.exists(path):
        pytest.skip("fixture pipe_full30_64.pt not generated")
    GC = full["GC"]
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    pc = GC.FULL_CASES["pipe_full30_64"]
    gold = torch.load(path)
    pin = pinning("pipe_full30_64")
    tok = StandInTokenizer()
    dim = full["configs"].UNET_VIDEO["cross_attention_dim"]
    clip = full["synth"].synth_clip(1, pc["t"], pc["h"], pc["w"], seed=pc["clip_seed"])
    unet = full["unet"]
    unet.stream_dtype = STREAMS[mode]
    res = {}
    import models_video.pipeline_upscale_a_video as PM
    real_randn = PM.randn_tensor

    def common_noise(shape, generator=None, device=None, dtype=None, **kw):      # as in make_golden.make_full30_golden
        return real_randn(shape, generator=generator, device=device, dtype=torch.float32, **kw).to(dtype)
    PM.randn_tensor = common_noise
    try:
        for draws in ("fp32", "half"):          # dtype of the text embeddings = dtype of the two noise tensors (pipeline:547,573)
            pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, dim, dtype=torch.float32 if draws == "fp32" else torch.float16),
                                        tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED),
                                        vae=full["vae"], unet=unet, propagator=None).to(dev)
            pipe.latents_trace = []
            out, lat = pipe(pc["prompt"], image=clip.to(dev), generator=torch.Generator().manual_seed(10),
                            num_inference_steps=pc["steps"], guidance_scale=pc["guidance"], noise_level=pc["noise_level"],
                            negative_prompt=pc["negative"], return_dict=False)
            res[draws] = dict(trace=[x.float().cpu() for x in pipe.latents_trace], img=out.float().cpu())
    finally:
        unet.stream_dtype = None
        PM.randn_tensor = real_randn
    steps = gold["steps"]
    curve32 = [rel_l2(res["fp32"]["trace"][k - 1], gold["latents_fp32"][i]) for i, k in enumerate(steps)]
    curve16 = [rel_l2(res["half"]["trace"][k - 1], gold["latents_half"][i]) for i, k in enumerate(steps)]
    curve16v32 = [rel_l2(res["half"]["trace"][k - 1], gold["latents_fp32"][i]) for i, k in enumerate(steps)]
    ref_noise = list(pin["reference_half_vs_fp32_latents_rel_l2_at_kept_steps"])
    g32 = gold["images_fp32_sub2"].float()
    unsat = g32.abs() < 0.999
    e_img = rel_l2(res["fp32"]["img"][..., ::2, ::2][unsat], g32[unsat])
    report("r3_pipe_full30_64_" + mode, steps=list(steps), engine_fp32_draws_vs_reference_fp32=curve32,
           engine_half_draws_vs_reference_half_run=curve16, engine_half_draws_vs_reference_fp32=curve16v32,
           reference_half_vs_its_fp32=ref_noise, image_rel_l2_unsaturated_vs_reference_fp32=e_img)
    # measured (MI355X, round 3): fp32 stream 1.84e-4 after 5 steps, 8.6e-4 after 30 (image 1.37e-3) — the bar
    # at step 30 is BASELINE.json's stated 1e-3, the others measured + 25 %; fp16 rows 6.6e-4 / 2.21e-3 (3.0e-3); the
    # reference's own half pipeline vs its fp32 run: 1.12e-3 / 3.0e-3 (4.0e-3)
    i5, i30 = list(steps).index(5), list(steps).index(30)
    bars = {"fp32_stream": (2.3e-4, 1.0e-3, 1.7e-3), "fp16_stream": (8.3e-4, 2.8e-3, 3.8e-3)}[mode]
    assert curve32[i5] < bars[0] and curve32[i30] < bars[1] and e_img < bars[2], (curve32, e_img)
    assert curve32[i30] < ref_noise[i30]                # closer to the fp32 trajectory than the reference's half pipeline
    # in the CLI's mix (fp16 noise tensors) the engine and the reference half run are two fp16-grade evaluations of the
    # same trajectory: their distance stays within 1.25x of the reference's own half-vs-fp32 distance
    assert curve16[i30] < 1.25 * ref_noise[i30], (curve16, ref_noise)


def test_vaevideo_full_width_decode_vs_reference(full, dev):
    """55 M-parameter `vae_video` decoder (BASELINE configs[4]: 3x3x3 convs in every decoder ResNet, LR frames through
    condition_in + the SFT fuse block) at the released width: one 3-frame chunk 48x48 -> 192x192 WITH the LR conditioning,
    against the reference module's fp32 decode (tests/golden/vaevideo_full_t3_48.pt)."""
    GC = full["GC"]
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    vae = AutoencoderKLVideo.from_config(dict(full["configs"].VAE_VIDEO))
    vae.load_state_dict(full["synth"].synth_state_dict(vae.state_dict(), seed=4321), strict=True)
    vae = vae.to(dev).eval()
    _, t, h, w = GC.FULL_CASES["vaevideo_full_t3_48"]
    z, img = GC.vae_inputs(1, t, h, w)
    with torch.no_grad():
        out = vae.decode(z.to(dev), img.to(dev), 1.0).sample
        out0 = vae.decode(z.to(dev), img.to(dev), 0.0).sample
    gold = torch.load(os.path.join(GOLD, "vaevideo_full_t3_48.pt"))
    e = rel_l2(out, gold)
    report("r3_vaevideo_full_t3_48", rel_l2_vs_reference_fp32=e, ref_absmean=pinning("vaevideo_full_t3_48")["ref_absmean"])
    assert out.shape == gold.shape and out.dtype == torch.float32
    assert e < 1e-3, e
    assert rel_l2(out0, gold) > 0.5                     # the LR conditioning weight does change the result
