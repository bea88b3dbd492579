"""Host side of the engine on a box WITHOUT a GPU: the product modules (`models_video.UNetVideoModel`,
`AutoencoderKLVideo`, `VideoUpscalePipeline`) run with the `uav.ops` entry points swapped for the torch stand-ins of
tests/cpu_ops.py, and are compared with the oracle / the reference fixtures.  What this covers is everything that is NOT
a kernel: weight packing (`pack_conv`, fused q|k|v rows, GEGLU interleave), the module orchestration (skip stack, two
source pointers instead of concatenation, forced upsample size, CFG-shared head, text K/V cache), layout conversion and
the pipeline loops.  The kernels themselves are covered by the `-m gpu` tests through the C ABI.

Tolerance: the stand-ins round every stored tensor to fp16 like the kernels do, so the distance to the fp32 oracle is
the same fp16 noise as on the GPU (2.2e-3 there): asserted <= the reference's own fp16 deviation (PINNING.json).
"""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cpu_ops  # noqa: E402
import golden_cases as GC  # noqa: E402
import synth  # noqa: E402
import uav_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.fixture()
def cpu_engine():
    cpu_ops.install()
    try:
        yield
    finally:
        cpu_ops.restore()


@pytest.fixture(scope="module")
def unet_and_sd():
    from models_video.unet_video import UNetVideoModel
    unet = UNetVideoModel.from_config(dict(GC.UNET_TINY))
    usd = synth.synth_state_dict(unet.state_dict(), seed=1234)
    unet.load_state_dict(usd, strict=True)
    return unet.eval(), usd


@pytest.mark.parametrize("name,shape", [("unet_t4_16", (2, 4, 16, 16)), ("unet_t3_20x28", (2, 3, 20, 28))])
def test_unet_forward_host_side(cpu_engine, unet_and_sd, name, shape):
    unet, usd = unet_and_sd
    sample, low, ehs, ts, cl = GC.unet_inputs(*shape, GC.UNET_TINY["cross_attention_dim"])
    with torch.no_grad():
        out = unet(sample.half(), ts, low.half(), encoder_hidden_states=ehs.half(), class_labels=cl).sample
    gold = torch.load(os.path.join(GOLD, name + ".pt"))
    noise = json.load(open(os.path.join(GOLD, "PINNING.json")))["cases"][name]["reference_fp16_vs_fp32_rel_l2"]
    assert out.shape == gold.shape and out.dtype == torch.float16
    assert rel_l2(out, gold) <= noise, (rel_l2(out, gold), noise)


def test_unet_forward_host_side_fp32_stream(cpu_engine, unet_and_sd):
    """UNetVideoModel.stream_dtype = float32 (fp32 residual stream, fp16 MFMA operands): the host orchestration — which
    tensors are fp32 (conv outputs, skip stack, token stream, GroupNorm / LayerNorm inputs), which are fp16 operands (norm
    outputs, the raw copy for the shortcut conv, block tails feeding proj_out / shift_conv) — against the reference
    fixture; clearly below the fp16-row mode and below the reference's own fp16 noise."""
    unet, usd = unet_and_sd
    name, shape = "unet_t3_20x28", (2, 3, 20, 28)
    sample, low, ehs, ts, cl = GC.unet_inputs(*shape, GC.UNET_TINY["cross_attention_dim"])
    gold = torch.load(os.path.join(GOLD, name + ".pt"))
    noise = json.load(open(os.path.join(GOLD, "PINNING.json")))["cases"][name]["reference_fp16_vs_fp32_rel_l2"]
    errs = {}
    try:
        for mode, dt in (("f16", torch.float16), ("f32", torch.float32)):
            unet.stream_dtype = dt
            with torch.no_grad():
                out = unet(sample, ts, low, encoder_hidden_states=ehs.half(), class_labels=cl).sample      # fp32 in -> fp32 out
            assert out.dtype == torch.float32
            errs[mode] = rel_l2(out, gold)
    finally:
        unet.stream_dtype = None
    assert errs["f32"] < 0.75 * errs["f16"] and errs["f32"] < 0.7 * noise, (errs, noise)


def test_unet_cfg_shared_head_host_side(cpu_engine, unet_and_sd):
    unet, _ = unet_and_sd
    sample, low, ehs, ts, cl = GC.unet_inputs(1, 3, 16, 16, GC.UNET_TINY["cross_attention_dim"])
    g = torch.Generator().manual_seed(11)
    ehs2 = torch.cat([ehs, torch.randn(ehs.shape, generator=g)]).half()
    s2, l2 = torch.cat([sample] * 2).half(), torch.cat([low] * 2).half()
    with torch.no_grad():
        a = unet(s2, ts, l2, encoder_hidden_states=ehs2, class_labels=cl).sample
        b = unet(s2, ts, l2, encoder_hidden_states=ehs2, class_labels=cl, cfg_shared_input=True).sample
    # bit-identical on the GPU (tests/test_models_gpu.py).  ATen's CPU convs block batch 1 and batch 2 differently: a
    # last-bit fp32 difference flips a few fp16 roundings in the head and those grow to the usual fp16 noise level
    assert rel_l2(a, b) < 5e-3 and rel_l2(a[0], a[1]) > 2e-2


@pytest.mark.parametrize("name,cfg", [("vae3d", GC.VAE3D_TINY), ("vaevideo", GC.VAEVIDEO_TINY)])
def test_vae_decode_host_side(cpu_engine, name, cfg):
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    vae = AutoencoderKLVideo.from_config(dict(cfg))
    vsd = synth.synth_state_dict(vae.state_dict(), seed=4321)
    vae.load_state_dict(vsd, strict=True)
    z, img = GC.vae_inputs(1, 3, 16, 16)
    with torch.no_grad():
        out = vae.eval().decode(z, img, 1.0).sample
    gold = torch.load(os.path.join(GOLD, name + "_t3_16.pt"))
    assert out.shape == gold.shape and out.dtype == torch.float32
    assert rel_l2(out, gold) < 5e-3


def test_pipeline_end_to_end_host_side(cpu_engine, unet_and_sd):
    """The whole product stack (pipeline + UNetVideoModel + AutoencoderKLVideo + schedulers) on the CPU stand-ins
    against the reference pipeline's fixture pipe_t8_vae3d (3 DDIM steps, guidance 6)."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    unet, _ = unet_and_sd
    case = GC.PIPE_CASES["pipe_t8_vae3d"]
    vae = AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY))
    vae.load_state_dict(synth.synth_state_dict(vae.state_dict(), seed=4321), strict=True)
    tok = StandInTokenizer()
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, GC.UNET_TINY["cross_attention_dim"], dtype=torch.float32),
                                tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED),
                                vae=vae.eval(), unet=unet, propagator=None).to("cpu")
    image, _ = GC.pipeline_inputs(case)
    out, lat = pipe(case["prompt"], image=image, generator=torch.Generator().manual_seed(10), num_inference_steps=case["steps"],
                    guidance_scale=case["guidance"], noise_level=case["noise_level"], negative_prompt=case["negative"],
                    return_dict=False)
    gold = torch.load(os.path.join(GOLD, "pipe_t8_vae3d.pt"))
    assert out.shape == gold["images"].shape and out.dtype == torch.float32 and float(out.abs().max()) <= 1.0
    assert rel_l2(lat, gold["latents"]) < 1e-2                      # same bar as the GPU test
    unsat = gold["images"].float().abs() < 0.999
    assert rel_l2(out[unsat], gold["images"].float()[unsat]) < 3e-2


def test_vae_decode_w_lr_host_side(cpu_engine):
    """w_lr = 0.5 through the video VAE's SFT conditioning: oracle and product module vs the reference fixture."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    vae = AutoencoderKLVideo.from_config(dict(GC.VAEVIDEO_TINY))
    vsd = synth.synth_state_dict(vae.state_dict(), seed=4321)
    vae.load_state_dict(vsd, strict=True)
    z, img = GC.vae_inputs(1, 3, 16, 16)
    gold = torch.load(os.path.join(GOLD, "vaevideo_t3_16_wlr05.pt"))
    with torch.no_grad():
        assert rel_l2(O.vae_decode(vsd, GC.VAEVIDEO_TINY, z, img, 0.5), gold) < 1e-3          # fixture stored in fp16
        out = vae.eval().decode(z, img, 0.5).sample
        full = vae.decode(z, img, 1.0).sample
    assert rel_l2(out, gold) < 5e-3
    assert rel_l2(full, gold) > 1e-2                                                           # the weight does matter


def test_pipeline_duplicate_tail_window_host_side(cpu_engine, unet_and_sd):
    """T = 14 (windows [0,8), [6,14), [6,14) again) through the whole product stack vs the reference pipeline's latents."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    unet, _ = unet_and_sd
    vae = AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY))
    vae.load_state_dict(synth.synth_state_dict(vae.state_dict(), seed=4321), strict=True)
    tok = StandInTokenizer()
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, GC.UNET_TINY["cross_attention_dim"], dtype=torch.float32),
                                tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED),
                                vae=vae.eval(), unet=unet, propagator=None).to("cpu")
    clip = synth.synth_clip(1, 14, 16, 16, seed=14)
    out, lat = pipe("p", image=clip, generator=torch.Generator().manual_seed(10), num_inference_steps=2, guidance_scale=6.0,
                    noise_level=120, negative_prompt="n", return_dict=False)
    gold = torch.load(os.path.join(GOLD, "pipe_t14_dup_tail.pt"))
    assert out.shape == (1, 3, 14, 64, 64)
    assert rel_l2(lat, gold["latents"]) < 1e-2


def test_tiled_clip_host_side(cpu_engine, unet_and_sd):
    """uav.tiling.upscale_tiled over the whole product stack (H = 68: forced-upsample-size path in every tile) vs the
    fixture made by the reference CLI loop around the reference pipeline."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav import tiling
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    unet, _ = unet_and_sd
    vae = AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY))
    vae.load_state_dict(synth.synth_state_dict(vae.state_dict(), seed=4321), strict=True)
    tok = StandInTokenizer()
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, GC.UNET_TINY["cross_attention_dim"], dtype=torch.float32),
                                tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED),
                                vae=vae.eval(), unet=unet, propagator=None).to("cpu")
    clip = synth.synth_clip(1, 2, 68, 160, seed=33)
    out = tiling.upscale_tiled(pipe, "p", clip, None, torch.Generator().manual_seed(10), tile_size=64, num_inference_steps=2,
                               guidance_scale=6.0, noise_level=120, negative_prompt="n")
    gold = torch.load(os.path.join(GOLD, "pipe_tiled_t2_68x160.pt"))
    for mine, ref in ((out[..., ::4, ::4], gold["sub4"].float()), (out[..., :, 240:272], gold["seam"].float())):
        unsat = ref.abs() < 0.999
        assert rel_l2(mine[unsat], ref[unsat]) < 3e-2


def test_pipeline_with_propagation_host_side(cpu_engine, unet_and_sd):
    """BASELINE config-3 flow through the whole product stack incl. the real Propagation module (two sweeps, in-place
    frame views) and the video VAE, T = 10 (two windows): vs the reference pipeline's fixture."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.propagation_module import Propagation
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    unet, _ = unet_and_sd
    case = GC.PIPE_CASES["pipe_t10_vaevideo_prop"]
    vae = AutoencoderKLVideo.from_config(dict(GC.VAEVIDEO_TINY))
    vae.load_state_dict(synth.synth_state_dict(vae.state_dict(), seed=4321), strict=True)
    tok = StandInTokenizer()
    prop = Propagation(4, learnable=False)
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, GC.UNET_TINY["cross_attention_dim"], dtype=torch.float32),
                                tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED),
                                vae=vae.eval(), unet=unet, propagator=prop).to("cpu")
    image, flows = GC.pipeline_inputs(case)
    out, lat = pipe(case["prompt"], image=image, flows_bi=flows, generator=torch.Generator().manual_seed(10),
                    num_inference_steps=case["steps"], guidance_scale=case["guidance"], noise_level=case["noise_level"],
                    negative_prompt=case["negative"], propagation_steps=list(case["propagation_steps"]), return_dict=False)
    gold = torch.load(os.path.join(GOLD, "pipe_t10_vaevideo_prop.pt"))
    assert rel_l2(lat, gold["latents"]) < 1e-2
    unsat = gold["images"].float().abs() < 0.999
    assert rel_l2(out[unsat], gold["images"].float()[unsat]) < 3e-2


@pytest.mark.parametrize("interp", ["nearest", "bilinear"])
def test_propagation_module_host_side(cpu_engine, interp):
    from models_video.propagation_module import Propagation
    x, ff, fb = GC.prop_inputs(8, 24, 32)
    out = Propagation(4, learnable=False)(x.half(), ff.half(), fb.half(), interpolation=interp, mode="fuse", fuse_scale=0.5,
                                          alpha1=0.001, alpha2=0.05)
    gold = torch.load(os.path.join(GOLD, f"propagation_{interp}.pt"))
    assert rel_l2(out, gold) < 5e-3


def test_packed_cache_invalidation_rules(cpu_engine):
    """what the packed-weight stamp sees and what it does not.  `p.copy_` (in-place on the Parameter) bumps
    `p._version` -> repacked automatically; `p.data.copy_` writes through a detached alias with its own version counter ->
    invisible, the documented remedy is `engine.invalidate_packed(model)` (also what `init_weights.random_init_` calls)."""
    from models_video.resnet import InflatedConv3d
    from uav import engine as E
    conv = InflatedConv3d(64, 64, 1).eval()
    x = torch.randn(1, 64, 1, 4, 4)
    with torch.no_grad():
        y0 = conv(x)
        w_new = torch.randn_like(conv.weight)
        conv.weight.copy_(w_new)                                   # visible: version counter of the Parameter moves
        y1 = conv(x)
        assert rel_l2(y1, torch.nn.functional.conv2d(x[:, :, 0], w_new, conv.bias)[:, :, None]) < 2e-3
        assert rel_l2(y1, y0) > 0.5
        w_new2 = torch.randn_like(conv.weight)
        v = conv.weight._version
        conv.weight.data.copy_(w_new2)                             # invisible to the stamp ...
        assert conv.weight._version == v
        y_stale = conv(x)
        assert torch.equal(y_stale, y1)                            # ... so the packed copy of the old weights is served
        E.invalidate_packed(conv)                                  # the documented remedy
        y2 = conv(x)
        assert rel_l2(y2, torch.nn.functional.conv2d(x[:, :, 0], w_new2, conv.bias)[:, :, None]) < 2e-3


def test_oracle_under_gpu_shim_equals_oracle():
    """oracle/gpu_shim.py (the GEMM-per-tap convolutions that let the oracle run on the GPU without MIOpen) on the CPU:
    the tiny UNet and both tiny VAE decoders give the same result as the plain oracle (fp32 summation order only)."""
    import gpu_shim
    from models_video.unet_video import UNetVideoModel
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    usd = synth.synth_state_dict(UNetVideoModel.from_config(dict(GC.UNET_TINY)).state_dict(), seed=1234)
    sample, low, ehs, ts, cl = GC.unet_inputs(2, 3, 20, 28, GC.UNET_TINY["cross_attention_dim"])
    with torch.no_grad():
        a = O.unet_forward(usd, GC.UNET_TINY, sample, ts, low, ehs, cl)
    with gpu_shim.oracle_on("cpu") as OS:
        b = OS.unet_forward(usd, GC.UNET_TINY, sample, ts, low, ehs, cl)
    assert rel_l2(b, a) < 1e-5
    assert rel_l2(a, torch.load(os.path.join(GOLD, "unet_t3_20x28.pt"))) < 1e-3        # fixture stored in fp16
    for cfg in (GC.VAE3D_TINY, GC.VAEVIDEO_TINY):
        vsd = synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(cfg)).state_dict(), seed=4321)
        z, img = GC.vae_inputs(1, 3, 16, 16)
        with torch.no_grad():
            a = O.vae_decode(vsd, cfg, z, img, 1.0)
        with gpu_shim.oracle_on("cpu") as OS:
            b = OS.vae_decode(vsd, cfg, z, img, 1.0)
        assert rel_l2(b, a) < 1e-5
    assert O.F is torch.nn.functional                               # the shim is gone again


def test_pack_conv_with_shortcut_layout(cpu_engine):
    """`conv_shortcut(x) + conv2(h)` packed as ONE weight matrix (ops.pack_conv_with_shortcut): K = taps x (C + Cs), the 1x1
    weights at the centre tap of the extra channels, zeros elsewhere, bias = b2 + bs, `k_logical` = 9 C + Cs for the FLOP
    accounting; with the [hi | lo] operand pair the shortcut weights are repeated.  Checked through the conv stand-in (which
    multiplies the structural zeros like the 128x128 kernel does) against the two convs."""
    from uav import ops
    g = torch.Generator().manual_seed(8)
    c, cs, o, h, w, n_img = 64, 128, 64, 6, 5, 2
    w2 = (torch.randn(o, c, 3, 3, generator=g) * 0.05).half().float(); b2 = torch.randn(o, generator=g)
    ws = (torch.randn(o, cs, 1, 1, generator=g) * 0.1).half().float(); bs = torch.randn(o, generator=g)
    hh = torch.randn(n_img * h * w, c, generator=g).half()
    x32 = torch.randn(n_img * h * w, cs, generator=g) * 3
    hi = x32.half(); lo = (x32 - hi.float()).half()
    for rep, raw in ((1, hi), (2, torch.cat([hi, lo], dim=1))):
        cw = ops.pack_conv_with_shortcut(w2, b2, ws, bs, rep)
        assert cw.cin_p == c + rep * cs and cw.k_logical == 9 * c + cs and cw.kh == cw.kw == 3
        wk = cw.w[:o, : 9 * (c + rep * cs)].float().reshape(o, 9, c + rep * cs)
        assert torch.equal(wk[:, :, :c].permute(0, 2, 1).reshape(o, c, 3, 3), w2)
        assert torch.count_nonzero(wk[:, [0, 1, 2, 3, 5, 6, 7, 8], c:]) == 0                 # off-centre taps of the shortcut channels
        assert torch.equal(wk[:, 4, c:c + cs], ws.reshape(o, cs)) and (rep == 1 or torch.equal(wk[:, 4, c + cs:], ws.reshape(o, cs)))
        assert torch.allclose(cw.bias[:o], b2 + bs)
        y = ops.conv_gemm(hh, cw, a2=raw, a2_center=True, n_img=n_img, t_len=1, hi=h, wi=w, out_f32=True)
        xs = raw[:, :cs].float() + (raw[:, cs:].float() if rep == 2 else 0)
        ref = torch.nn.functional.conv2d(hh.float().reshape(n_img, h, w, c).permute(0, 3, 1, 2), w2, b2, padding=1) + \
            torch.nn.functional.conv2d(xs.reshape(n_img, h, w, cs).permute(0, 3, 1, 2), ws, bs)
        assert rel_l2(y, ref.permute(0, 2, 3, 1).reshape(-1, o)) < 1e-5
    assert rel_l2(hi.float() + lo.float(), x32) < 1e-6                                         # the pair carries x to ~22 bits


def test_stream_set_ordering_rules(monkeypatch):
    """uav/streams.py without a GPU: the ORDER of waits and launches StreamSet.map issues (fake stream objects record it).  Every
    side stream waits for the caller's stream before its first unit, unit k goes to side stream k % n, the caller's stream waits
    for every side stream that was used before map returns, and a single unit never leaves the caller's stream."""
    import contextlib
    from uav import streams
    log = []

    class FakeStream:
        def __init__(self, name="side", device=None):
            self.name = name if name != "side" else f"s{len([e for e in log if e[0] == 'new'])}"
            self.cuda_stream = id(self)
            log.append(("new", self.name))

        def wait_stream(self, other):
            log.append(("wait", self.name, other.name))

    cur = FakeStream("cur")
    state = {"current": cur}

    @contextlib.contextmanager
    def fake_ctx(st):
        prev, state["current"] = state["current"], st
        try:
            yield
        finally:
            state["current"] = prev
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream(device=device))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: cur)
    monkeypatch.setattr(torch.cuda, "stream", fake_ctx)
    ss = streams.StreamSet("cuda:0", 2)
    log.clear()
    out = ss.map([10, 11, 12], lambda it: (log.append(("unit", it, state["current"].name)), it * 2)[1])
    assert out == [20, 22, 24]
    assert log == [("wait", "s1", "cur"), ("unit", 10, "s1"), ("wait", "s2", "cur"), ("unit", 11, "s2"), ("unit", 12, "s1"),
                   ("wait", "cur", "s1"), ("wait", "cur", "s2")]
    log.clear()
    assert ss.map([7], lambda it: (log.append(("unit", it, state["current"].name)), it)[1]) == [7]
    assert log == [("unit", 7, "cur")]                       # one unit: stays on the caller's stream, no waits
    log.clear()
    assert ss.map([], lambda it: it) == [] and log == []
    # one StreamSet per (device, n, calling stream): two host threads on their own streams do not share side streams
    monkeypatch.setattr(streams, "_SETS", {})
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    a = streams.stream_set("cuda:0", 2)
    assert streams.stream_set("cuda:0", 2) is a and streams.stream_set("cuda:0", 3) is not a
    other = FakeStream("cur2")
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: other)
    assert streams.stream_set("cuda:0", 2) is not a


def test_pipeline_overlap_units_host_side(cpu_engine, unet_and_sd, monkeypatch):
    """Which units the pipeline hands to the stream set (`_stream_set`, `overlap_streams`, `overlap_split_cfg`) — on CPU with a
    recording stand-in for uav.streams.StreamSet that evaluates serially: T = 14 gives its 2 unique windows per DDIM step and 5
    decode chunks; an 8-frame clip stays serial unless the guidance branches are asked for, which gives 2 batch-1 units per
    step (the shard_cfg decomposition) and 3 decode chunks.  Results equal the serial call's."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav import streams
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    unet, _ = unet_and_sd
    vae = AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY))
    vae.load_state_dict(synth.synth_state_dict(vae.state_dict(), seed=4321), strict=True)
    tok = StandInTokenizer()
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, GC.UNET_TINY["cross_attention_dim"], dtype=torch.float32),
                                tokenizer=tok, low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED),
                                vae=vae.eval(), unet=unet, propagator=None).to("cpu")
    calls = []

    class Recorder:
        streams = [0, 1]

        def map(self, items, fn):
            items = list(items)
            calls.append(items)
            return [fn(it) for it in items]
    # the decision itself (a torch.device("cuda") can be named without a GPU)
    monkeypatch.setattr(streams, "stream_set", lambda device, n: Recorder())
    cuda = torch.device("cuda", 0)
    assert pipe.overlap_streams == 2 and not pipe.overlap_split_cfg
    assert pipe._stream_set(cuda, 1, True) is None and pipe._stream_set(torch.device("cpu"), 3, True) is None
    assert isinstance(pipe._stream_set(cuda, 2, True), Recorder)
    pipe.overlap_split_cfg = True
    assert isinstance(pipe._stream_set(cuda, 1, True), Recorder) and pipe._stream_set(cuda, 1, False) is None
    pipe.shard_windows = True
    assert pipe._stream_set(cuda, 3, True) is None
    pipe.shard_windows, pipe.overlap_split_cfg, pipe.overlap_streams = False, False, 0
    assert pipe._stream_set(cuda, 3, True) is None
    pipe.overlap_streams = 2

    def run(t, overlapped, split=False):
        pipe.overlap_split_cfg = split
        clip = synth.synth_clip(1, t, 16, 16, seed=14)
        if overlapped:
            monkeypatch.setattr(pipe, "_stream_set", lambda device, n_windows, do_cfg: Recorder() if (n_windows > 1 or split) else None)
        else:
            monkeypatch.setattr(pipe, "_stream_set", lambda device, n_windows, do_cfg: None)
        calls.clear()
        return pipe("p", image=clip, generator=torch.Generator().manual_seed(10), num_inference_steps=2, guidance_scale=6.0,
                    noise_level=120, negative_prompt="n", return_dict=False)
    ref = run(14, False)
    assert calls == []
    got = run(14, True)
    assert calls == [[((0, 8), None), ((6, 14), None)]] * 2 + [[0, 3, 6, 9, 12]]
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    ref8 = run(8, False)
    assert run(8, True)[0].shape == ref8[0].shape and calls == []            # one window, no split: serial
    got8 = run(8, True, split=True)
    assert calls == [[((0, 8), 0), ((0, 8), 1)]] * 2 + [[0, 3, 6]]
    assert rel_l2(got8[1], ref8[1]) < 5e-3          # batch-1 units vs the batch-2 evaluation: ATen's CPU convs block differently (see above)


def test_transformer_block_fused_feed_forward_host_side(cpu_engine):
    """BasicTransformerBlock at the 512-channel width with the feed-forward as ONE launch (ops.ff_sublayer; weights re-packed into the
    kernel's fragment stream by ops.pack_ff_weights) against LayerNorm + the two GEMM launches: the host side — which blocks qualify, the
    stream layout (the stand-in un-packs it), norm3 not asked of the attention launch, the hi | lo pair for proj_out."""
    from uav import engine as E, ops
    from models_video.attention import BasicTransformerBlock
    g = torch.Generator().manual_seed(5)
    blk = BasicTransformerBlock(512, 8, 64, cross_attention_dim=64, only_cross_attention=True)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.2))
        for ln in (blk.norm1, blk.norm2, blk.norm_temporal, blk.norm3):
            ln.weight.add_(1.0)
    blk = blk.half().eval()
    geom = E.Geom(1, 8, 4, 4)                                   # 128 rows: one tile of every fused kernel
    x = torch.randn(geom.rows, 512, generator=g) * 1.2
    ehs = torch.randn(7, 64, generator=g).half()
    assert blk.ff.fused_params(x, blk.norm3) is not None and blk.ff.fused_params(x.half(), blk.norm3) is None
    assert blk.ff.fused_params(x[:100], blk.norm3) is None      # not whole 128-row tiles
    old = E.FF_FUSED
    res = {}
    try:
        for on in (True, False):
            E.FF_FUSED = on
            E.invalidate_packed(blk)
            with torch.no_grad():
                res[on] = (blk.run(x.clone(), geom, ehs, 7), blk.run(x.clone(), geom, ehs, 7, out_hilo=True))
            if on:
                assert ops.next_ln_of(res[on][0], blk.norm3.weight, blk.norm3.bias, blk.norm3.eps) is None
    finally:
        E.FF_FUSED = old
        E.invalidate_packed(blk)
    (y1, h1), (y0, h0) = res[True], res[False]
    assert y1.dtype == y0.dtype == torch.float32 and h1.dtype == h0.dtype == torch.float16 and h1.shape == (geom.rows, 1024)
    assert rel_l2(y1, y0) < 2e-4                                # the two forms on the CPU: fp32 summation order only
    assert torch.equal(h1, ops.cast_hilo(y1))
    # ... and the whole block as one launch on / off (ops.block_sublayers against the attention launch + the feed-forward launch)
    oldb = E.BLOCK_FF_FUSED
    try:
        E.BLOCK_FF_FUSED = False
        E.invalidate_packed(blk)
        with torch.no_grad():
            y2, h2 = blk.run(x.clone(), geom, ehs, 7), blk.run(x.clone(), geom, ehs, 7, out_hilo=True)
    finally:
        E.BLOCK_FF_FUSED = oldb
        E.invalidate_packed(blk)
    assert torch.equal(y2, y1) and torch.equal(h2, h1)          # the stand-ins compose the same arithmetic


def test_transformer3d_groupnorm_proj_in_inside_the_block_launch_host_side(cpu_engine):
    """Transformer3DModel at the 512-channel width: GroupNorm apply -> proj_in -> the whole block as ONE launch (ops.block_sublayers with
    proj_in) against the GroupNorm pass + proj_in launch + block launch — the host side: scale | shift rows per frame, proj_in re-packed as
    'out' fragments, the hi | lo pair handed to proj_out, the switch."""
    from uav import engine as E
    from models_video.attention import Transformer3DModel
    g = torch.Generator().manual_seed(9)
    m = Transformer3DModel(num_attention_heads=8, attention_head_dim=64, in_channels=512, num_layers=1, cross_attention_dim=64,
                           use_linear_projection=True, only_cross_attention=True)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.04 if p_.dim() > 1 else 0.2))
        for mod in m.modules():
            if isinstance(mod, (torch.nn.LayerNorm, torch.nn.GroupNorm)):
                mod.weight.add_(1.0)
    m = m.half().eval()
    geom = E.Geom(1, 8, 4, 4)
    x = torch.randn(geom.rows, 512, generator=g) * 1.5
    ehs = torch.randn(7, 64, generator=g).half()
    old = E.PROJ_IN_FUSED
    res = {}
    try:
        for on in (True, False):
            E.PROJ_IN_FUSED = on
            E.invalidate_packed(m)
            with torch.no_grad():
                res[on] = m.run(x.clone(), geom, ehs, 7)
    finally:
        E.PROJ_IN_FUSED = old
        E.invalidate_packed(m)
    assert res[True].dtype == res[False].dtype == torch.float32
    assert rel_l2(res[True], res[False]) < 2e-4, rel_l2(res[True], res[False])
