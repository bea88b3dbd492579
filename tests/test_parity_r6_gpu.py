"""Round-6 parity evidence on the GPU (VERDICT r5 missing #3 / next #1c): BASELINE configs[3] at its REAL length.

  * T = 32 frames, 320x320 -> 1280x1280, 30 DDIM steps, guidance 6, full width: the reference pipeline's window schedule
    (pipeline_upscale_a_video.py:601-635: 8-frame windows at stride 6 -> [0,8) [6,14) [12,20) [18,26) [24,32) + the duplicate
    tail window, running 0.5 / 0.5 epsilon blend on the overlaps) and the 3-frame decode chunks on the global frame index
    (:685-702, 11 chunks), against the fp32 oracle pipeline executed on the same GPU (oracle/gpu_shim.py; the oracle's window
    schedule is pinned against the reference pipeline's own run by tests/golden/pipe_full30_64_t14.pt).

The oracle side costs ~10 GPU-minutes (6 fp32 UNet evaluations per step), so the case only runs with UAV_PARITY_T32=1
(`tools/run.sh TAG t32`); the recorded run is profiles/r06_parity_configs3_t32_320.jsonl.  The engine runs the call twice — the
serial window loop and the windows on two HIP streams (`overlap_streams`, the one-GPU form of the window sharding) — and the
two must agree bit for bit."""
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

from test_parity_r4_gpu import GUIDANCE, NEGATIVE, NOISE_LEVEL, PROMPT, STEPS, build_models, image_errors, rel_l2, report  # noqa: E402

T32, H, W = 32, 320, 320


def _engine(dev, unet, vae, clip, overlap_streams):
    import golden_cases as GC
    from uav import configs
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    tok = StandInTokenizer()
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, configs.UNET_VIDEO["cross_attention_dim"], dtype=torch.float32),
                                tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED), vae=vae, unet=unet,
                                propagator=None).to(dev)
    pipe.overlap_streams = overlap_streams
    pipe.latents_trace = []
    torch.cuda.synchronize(); t0 = time.time()
    out, lat = pipe(PROMPT, image=clip.to(dev), generator=torch.Generator().manual_seed(10), num_inference_steps=STEPS,
                    guidance_scale=GUIDANCE, noise_level=NOISE_LEVEL, negative_prompt=NEGATIVE, return_dict=False)
    torch.cuda.synchronize()
    return dict(images=out, latents=lat, trace=pipe.latents_trace, seconds=time.time() - t0, mode=pipe.last_overlap_mode)


@pytest.mark.skipif(os.environ.get("UAV_PARITY_T32", "0") in ("", "0"), reason="~12 GPU-minutes (fp32 oracle of 6 windows x 30 steps): UAV_PARITY_T32=1")
def test_configs3_t32_window_schedule_at_320_vs_gpu_oracle(dev):
    import golden_cases as GC
    import gpu_shim
    import synth
    from uav import configs
    clip = synth.synth_clip(1, T32, H, W, seed=3, motion=(2, 1))
    unet, usd, vae, vsd = build_models(dev)
    eng = _engine(dev, unet, vae, clip, overlap_streams=0)
    eng2 = _engine(dev, unet, vae, clip, overlap_streams=2)
    same = bool(torch.equal(eng["images"], eng2["images"]) and torch.equal(eng["latents"], eng2["latents"]))
    eng_cpu = dict(images=eng["images"].float().cpu(), latents=eng["latents"].float().cpu(), trace=[x.float().cpu() for x in eng["trace"]])
    del eng2["images"], eng2["latents"], eng2["trace"]
    eng["images"] = eng["latents"] = eng["trace"] = None
    torch.cuda.empty_cache()
    gen = torch.Generator().manual_seed(10)
    lr_noise = torch.randn(clip.shape, generator=gen)
    lat0 = torch.randn((1, 4, T32, H, W), generator=gen)
    dim = configs.UNET_VIDEO["cross_attention_dim"]
    pe = torch.cat([synth.synth_prompt_embeds(NEGATIVE, dim), synth.synth_prompt_embeds(PROMPT, dim)])
    usd_d = {k: v.to(dev) for k, v in usd.items()}
    vsd_d = {k: v.to(dev) for k, v in vsd.items()}
    with gpu_shim.oracle_on(dev) as O:
        torch.cuda.synchronize(); t0 = time.time()
        img, lat, tr = O.pipeline_call(usd_d, configs.UNET_VIDEO, vsd_d, configs.VAE_3D, clip.to(dev), pe.to(dev), num_inference_steps=STEPS,
                                       guidance_scale=GUIDANCE, noise_level=NOISE_LEVEL, lr_noise=lr_noise.to(dev), latents=lat0.to(dev),
                                       scheduler_kwargs=GC.SCHED, return_trace=True)
        torch.cuda.synchronize(); osec = time.time() - t0
    curve = [rel_l2(e, o.cpu()) for e, o in zip(eng_cpu["trace"], tr)]
    e_lat = rel_l2(eng_cpu["latents"], lat.cpu())
    e_all, e_unsat, sat = image_errors(eng_cpu["images"], img.cpu())
    report("r6_configs3_t32_320x320_30steps_window_schedule_vs_gpu_oracle", shape=[T32, H, W], steps=STEPS, guidance=GUIDANCE,
           windows=O.window_schedule(T32), latents_rel_l2_per_step=curve, latents_rel_l2=e_lat, images_rel_l2_all_pixels=e_all,
           images_rel_l2_unsaturated=e_unsat, images_saturated_fraction=sat, engine_seconds_serial=eng["seconds"],
           engine_seconds_two_streams=eng2["seconds"], two_stream_mode=eng2["mode"], two_streams_bit_identical_to_serial=same,
           oracle_seconds=osec)
    assert len(curve) == STEPS and eng_cpu["images"].shape == (1, 3, T32, 4 * H, 4 * W)
    assert same, "windows on two streams must reproduce the serial window loop bit for bit"
    assert e_lat < 1.0e-3, (e_lat, curve)                 # the stated tolerance, like T = 8 and T = 14
    assert e_all < 1.0e-3, (e_all, e_unsat)
