"""CPU suite (`-m "not gpu"`): the oracle against the reference-generated golden vectors, host
logic of the drop-in package, state-dict compatibility with the reference, C-ABI export list,
no-CPU-fallback behaviour, and the world_size-2 gloo path of the multi-GPU helpers."""
import json
import os
import re
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")

import golden_cases as GC  # noqa: E402
import synth  # noqa: E402
import uav_oracle as O  # noqa: E402


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


# ---------------------------------------------------------------------------------------------
# 1. the oracle reproduces the fixtures the REFERENCE's own modules produced
def test_pinning_record():
    pin = json.load(open(os.path.join(GOLD, "PINNING.json")))
    for name, rec in pin["cases"].items():
        for k, v in rec.items():
            if "maxabs" in k:
                assert v < 1e-4, f"{name}: oracle vs reference {k} = {v}"


@pytest.fixture(scope="module")
def unet_sd():
    from models_video.unet_video import UNetVideoModel
    m = UNetVideoModel.from_config(dict(GC.UNET_TINY))
    return synth.synth_state_dict(m.state_dict(), seed=1234)


def RAFT_model():
    from models_video.RAFT.raft import RAFT
    return RAFT()


@pytest.mark.parametrize("name,shape", [("unet_t4_16", (2, 4, 16, 16)), ("unet_t3_20x28", (2, 3, 20, 28))])
def test_oracle_unet_vs_golden(unet_sd, name, shape):
    """The second case has H, W not multiples of 8: forced upsample size (reference unet_video.py:443-445,541-542)."""
    sample, low, ehs, ts, cl = GC.unet_inputs(*shape, GC.UNET_TINY["cross_attention_dim"])
    with torch.no_grad():
        out = O.unet_forward(unet_sd, GC.UNET_TINY, sample, ts, low, ehs, cl)
    gold = torch.load(os.path.join(GOLD, name + ".pt"))
    assert out.shape == gold.shape
    assert rel_l2(out, gold) < 1e-3          # fixture stored in fp16


def test_oracle_raft_bi_resize_path_vs_golden():
    """RAFT_bi on frames whose H, W are not multiples of 8 (pre-resize + flow resize with the reference's
    row-indexed rescale, raft_bi.py:11-16,49-53,62-63) against the reference-generated fixture."""
    m = RAFT_model()
    sd = synth.synth_state_dict(m.state_dict(), seed=777)
    clip = synth.synth_clip(1, 3, 132, 164, seed=5, motion=(2, 1))
    with torch.no_grad():
        ff, fb = O.raft_bi_forward(sd, clip, iters=3)
    gold = torch.load(os.path.join(GOLD, "raft_bi_t3_132x164.pt"))
    assert ff.shape == (1, 2, 2, 132, 164)
    assert rel_l2(ff, gold["forward"]) < 1e-3 and rel_l2(fb, gold["backward"]) < 1e-3     # fixture stored in fp16


@pytest.mark.parametrize("name,cfg", [("vae3d", GC.VAE3D_TINY), ("vaevideo", GC.VAEVIDEO_TINY)])
def test_oracle_vae_vs_golden(name, cfg):
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    vsd = synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(cfg)).state_dict(), seed=4321)
    z, img = GC.vae_inputs(1, 3, 16, 16)
    with torch.no_grad():
        out = O.vae_decode(vsd, cfg, z, img, 1.0)
    assert rel_l2(out, torch.load(os.path.join(GOLD, name + "_t3_16.pt"))) < 1e-3


def test_oracle_pipeline_vs_golden(unet_sd):
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from uav.standin_text import prompt_embedding
    case = GC.PIPE_CASES["pipe_t8_vae3d"]
    vsd = synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY)).state_dict(), seed=4321)
    image, flows = GC.pipeline_inputs(case)
    gen = torch.Generator().manual_seed(10)
    lr_noise = torch.randn(image.shape, generator=gen)
    lat0 = torch.randn((1, 4) + tuple(image.shape[2:]), generator=gen)
    dim = GC.UNET_TINY["cross_attention_dim"]
    pe = torch.cat([synth.synth_prompt_embeds(case["negative"], dim), synth.synth_prompt_embeds(case["prompt"], dim)])
    assert torch.equal(prompt_embedding(case["prompt"], dim), synth.synth_prompt_embeds(case["prompt"], dim))
    with torch.no_grad():
        img, lat = O.pipeline_call(unet_sd, GC.UNET_TINY, vsd, GC.VAE3D_TINY, image, pe, num_inference_steps=case["steps"],
                                   guidance_scale=case["guidance"], noise_level=case["noise_level"], lr_noise=lr_noise,
                                   latents=lat0, scheduler_kwargs=GC.SCHED)
    gold = torch.load(os.path.join(GOLD, "pipe_t8_vae3d.pt"))
    assert rel_l2(lat, gold["latents"]) < 1e-3
    assert rel_l2(img, gold["images"]) < 2e-3


def test_oracle_propagation_vs_golden():
    x, ff, fb = GC.prop_inputs(8, 24, 32)
    for interp in ("nearest", "bilinear"):
        out = O.propagation(x, ff, fb, interp, 0.5, 0.001, 0.05)
        gold = torch.load(os.path.join(GOLD, f"propagation_{interp}.pt")).float()
        assert (out - gold).abs().max().item() < 2e-3


def test_oracle_propagation_fp32_on_rounding_ties_vs_golden():
    """Round 4: fp32 latents are held to the reference's fp32 propagation run; inputs ON the .5 ties of the nearest warp
    (tests/golden/propagation_*_f32_*.pt, reference `Propagation` on CPU fp32 tensors).  The half-run fixture of the same
    inputs differs on ~half of the tie pixels, so the pair discriminates the two coordinate modes."""
    for name in ("propagation_nearest_f32_ties", "propagation_nearest_f32_wide", "propagation_bilinear_f32_wide"):
        kind, t, h, w, interp = GC.PROP_HALF_CASES[name.replace("_f32", "_half")]
        x, ff, fb = GC.prop_half_inputs(kind, t, h, w)
        out = O.propagation(x, ff, fb, interp, 0.5, 0.001, 0.05)
        gold = torch.load(os.path.join(GOLD, name + ".pt"))
        assert gold.dtype == torch.float32
        assert (out - gold).abs().max().item() < 1e-5, name
        half = torch.load(os.path.join(GOLD, name.replace("_f32", "_half") + ".pt")).float()
        assert ((half - gold).abs() > 1e-2).float().mean().item() > 0.1, name


def test_oracle_colorfix_vs_golden():
    """Oracle restatement of the colour fix against the reference's own outputs (tests/golden/colorfix.pt)."""
    lr, content = GC.colorfix_inputs()
    gold = torch.load(os.path.join(GOLD, "colorfix.pt"))
    style = O.bicubic4(lr)
    assert (style - gold["style_bicubic4"]).abs().max().item() < 1e-6
    assert (O.adain(content, style) - gold["adain"]).abs().max().item() < 1e-5
    assert (O.wavelet_reconstruction(content, style) - gold["wavelet"]).abs().max().item() < 1e-5
    high, low = O.wavelet_decomposition(content)
    assert (high - gold["high"]).abs().max().item() < 1e-5 and (low - gold["low"]).abs().max().item() < 1e-5


def test_bench_self_launch_refuses_without_enough_gpus(capsys):
    """`python bench.py --gpus N` with no launcher re-execs under torch.distributed.run (bench.self_launch); with fewer
    than N visible GPUs it must fail loudly instead of printing a 1-rank line."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("uav_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert torch.cuda.device_count() == 0
    assert bench.self_launch(2) == 2
    assert "only 0 GPU(s) are visible" in capsys.readouterr().err


def test_bench_help_renders():
    """argparse %-formats every help string: an unescaped `%` in one of them breaks `python bench.py --help` only."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    for flag in ("--gpus", "--steps", "--warmup", "--unet-stream", "--shard-windows", "--overlap-streams", "--clips-per-step"):
        assert flag in r.stdout


def test_oracle_ddim_vs_golden():
    rec = json.load(open(os.path.join(GOLD, "ddim.json")))
    sch = O.DDIM(**GC.SCHED)
    assert sch.set_timesteps(30) == rec["timesteps30"]
    assert rec["timesteps30"][:2] == [958, 925] and rec["timesteps30"][-2:] == [34, 1]
    for t, a in rec["alphas_cumprod_at"].items():
        assert abs(float(sch.alphas_cumprod[int(t)]) - a) < 1e-7


# ---------------------------------------------------------------------------------------------
# 2. drop-in surface: state-dict keys/shapes identical to the reference's modules
def _keys(model):
    return {k: list(v.shape) for k, v in model.state_dict().items()}


def test_state_dict_keys_match_reference():
    from uav import configs
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.unet_video import UNetVideoModel
    assert _keys(UNetVideoModel.from_config(dict(configs.UNET_VIDEO))) == json.load(open(os.path.join(GOLD, "unet_full_keys.json")))
    assert _keys(AutoencoderKLVideo.from_config(dict(configs.VAE_3D))) == json.load(open(os.path.join(GOLD, "vae3d_full_keys.json")))
    assert _keys(AutoencoderKLVideo.from_config(dict(configs.VAE_VIDEO))) == json.load(open(os.path.join(GOLD, "vaevideo_full_keys.json")))


def test_cli_import_surface():
    """The names inference_upscale_a_video.py imports (reference :39-45) exist in the drop-in package."""
    from models_video.RAFT.raft_bi import RAFT_bi  # noqa: F401
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo  # noqa: F401
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline  # noqa: F401
    from models_video.propagation_module import Propagation  # noqa: F401
    from models_video.scheduling_ddim import DDIMScheduler  # noqa: F401
    from models_video.unet_video import UNetVideoModel  # noqa: F401
    from models_video.color_correction import adaptive_instance_normalization, wavelet_reconstruction  # noqa: F401


def test_no_cpu_fallback():
    from uav import UavError
    from models_video.unet_video import UNetVideoModel
    unet = UNetVideoModel.from_config(dict(GC.UNET_TINY))
    with pytest.raises(UavError):
        unet(torch.zeros(2, 4, 4, 16, 16), 925, torch.zeros(2, 3, 4, 16, 16), encoder_hidden_states=torch.zeros(2, 77, 64),
             class_labels=torch.tensor([120]))


def test_pipeline_input_errors():
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    pipe = VideoUpscalePipeline(max_noise_level=350)
    img = torch.zeros(1, 3, 8, 16, 16)
    with pytest.raises(ValueError):
        pipe.check_inputs("a", img, 351)
    with pytest.raises(ValueError):
        pipe.check_inputs(None, img, 20)
    with pytest.raises(ValueError):
        pipe.check_inputs("a", img, 20, prompt_embeds=torch.zeros(1, 77, 8))
    with pytest.raises(ValueError):
        pipe.check_inputs(["a", "b"], img, 20)
    with pytest.raises(TypeError):
        pipe.check_inputs(None, img, 20, prompt_embeds=torch.zeros(1, 77, 8))   # the reference's len(None) crash


# ---------------------------------------------------------------------------------------------
# 3. C ABI
def test_cabi_exports_every_declared_symbol():
    from uav import _lib
    hdr = open(os.path.join(ROOT, "include", "uav_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(uav_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/uav_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.uav_version() == _lib.EXPECTED_ABI == int(re.search(r"#define UAV_ABI_VERSION (\d+)", hdr).group(1))


# ---------------------------------------------------------------------------------------------
# 4. host logic
def emulate_conv_gemm(rows, cw, n_img, t_len, hi, wi, stride, pad, upsample, out_hw=None):
    """Executable spec of uav_conv_gemm_f16's addressing (csrc/conv_gemm.hip src_pixel + k order)."""
    pt, ph, pw = pad
    if upsample:
        ho, wo = 2 * hi, 2 * wi
    else:
        ho, wo = (hi + 2 * ph - cw.kh) // stride + 1, (wi + 2 * pw - cw.kw) // stride + 1
    if out_hw:
        ho, wo = out_hw
    w = cw.w.float()[: cw.n]
    x = rows.float().reshape(n_img, hi, wi, cw.cin_p)
    out = torch.zeros(n_img, ho, wo, cw.n)
    for img in range(n_img):
        tloc = img % t_len
        for dt in range(cw.kt):
            tt = tloc + dt - pt
            if not (0 <= tt < t_len):
                continue
            for dy in range(cw.kh):
                for dx in range(cw.kw):
                    tap = (dt * cw.kh + dy) * cw.kw + dx
                    wk = w[:, tap * cw.cin_p:(tap + 1) * cw.cin_p]
                    for yo in range(ho):
                        for xo in range(wo):
                            if upsample:
                                yv, xv = yo + dy - ph, xo + dx - pw
                                if not (0 <= yv < ho and 0 <= xv < wo):
                                    continue
                                yi, xi = yv >> 1, xv >> 1
                            else:
                                yi, xi = yo * stride + dy - ph, xo * stride + dx - pw
                                if not (0 <= yi < hi and 0 <= xi < wi):
                                    continue
                            out[img, yo, xo] += wk @ x[img + dt - pt, yi, xi]
    return out + (cw.bias[: cw.n] if cw.bias is not None else 0)


@pytest.mark.parametrize("cin,cout,k3,stride,ups,t_len", [(64, 8, (1, 3, 3), 1, False, 1), (64, 8, (1, 3, 3), 2, False, 1),
                                                          (64, 4, (1, 3, 3), 1, True, 1), (64, 8, (3, 1, 1), 1, False, 3),
                                                          (7, 8, (1, 3, 3), 1, False, 1), (3, 4, (3, 3, 3), 1, False, 3)])
def test_pack_conv_and_kernel_addressing_spec(cin, cout, k3, stride, ups, t_len):
    from uav import ops
    g = torch.Generator().manual_seed(cin + cout)
    h, w_, nb = 5, 6, 1
    x5 = torch.randn(nb, cin, t_len, h, w_, generator=g)
    wt = torch.randn(cout, cin, *k3, generator=g).half().float()
    bias = torch.randn(cout, generator=g)
    ref = F.conv3d(F.interpolate(x5, scale_factor=[1.0, 2.0, 2.0], mode="nearest") if ups else x5, wt, bias,
                   stride=(1, stride, stride), padding=tuple(k // 2 for k in k3))
    cw = ops.pack_conv(wt, bias, device="cpu")
    rows = x5.permute(0, 2, 3, 4, 1).reshape(-1, cin)
    if cin % 64:
        rows = F.pad(rows, (0, 8 - cin))
    out = emulate_conv_gemm(rows, cw, nb * t_len, t_len, h, w_, stride, tuple(k // 2 for k in k3), ups)
    ho, wo = ref.shape[-2:]
    out5 = out[..., :cout].reshape(nb, t_len, ho, wo, cout).permute(0, 4, 1, 2, 3)
    assert (out5 - ref).abs().max().item() < 1e-3
    assert cw.n_pad % 128 == 0 and cw.k_pad % 64 == 0


def test_prompt_embedding_cache():
    """one text-encoder pass per distinct (prompt, negative prompt), not one per pipeline call / tile."""
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    tok = StandInTokenizer()
    enc = StandInTextEncoder(tok, 64, dtype=torch.float32)
    calls = []
    orig = enc.forward
    enc.forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    pipe = VideoUpscalePipeline(text_encoder=enc, tokenizer=tok)
    a = pipe._cached_prompt_embeds("a city street", "cpu", 1, True, "blur")
    b = pipe._cached_prompt_embeds("a city street", "cpu", 1, True, "blur")
    assert a is b and a.shape == (2, 77, 64) and len(calls) == 2              # prompt + negative prompt, once
    c = pipe._cached_prompt_embeds("a forest", "cpu", 1, True, "blur")
    assert c is not a and not torch.equal(c, a) and len(calls) == 4
    assert torch.equal(a, pipe._encode_prompt("a city street", "cpu", 1, True, "blur"))
    pipe.cache_prompt_embeds = False
    assert pipe._cached_prompt_embeds("a city street", "cpu", 1, True, "blur") is not a
    # tensors passed in by the caller are never cached
    pipe.cache_prompt_embeds = True
    pe = torch.zeros(1, 77, 64)
    n = len(pipe._prompt_cache)
    pipe._cached_prompt_embeds(None, "cpu", 1, False, None, prompt_embeds=pe)
    assert len(pipe._prompt_cache) == n


def test_tile_order_maps_are_bijections():
    """Python mirror of the launch-order maps in csrc (uav_common.h:xcd_remap, conv_gemm.hip SETUP_TILE frame-fastest
    order for temporal convs, persistent tile walk): every tile is produced exactly once."""
    def xcd_remap(bid, nwg):
        nx = 8
        q, r = divmod(nwg, nx)
        xcd, idx = bid % nx, bid // nx
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        return base + idx

    for nwg in (1, 7, 8, 9, 255, 256, 257, 1000, 12800):
        assert sorted(xcd_remap(b, nwg) for b in range(nwg)) == list(range(nwg))
        # consecutive remapped ids sit on one XCD (hardware places block b on XCD b % 8)
        by_xcd = {}
        for b in range(nwg):
            by_xcd.setdefault(b % 8, []).append(xcd_remap(b, nwg))
        for ids in by_xcd.values():
            assert ids == list(range(ids[0], ids[0] + len(ids)))

    def temporal_order(mt, hw, t_len, lm=256):
        if hw % lm:
            return mt
        s_ = hw // lm
        per_clip = s_ * t_len
        c, r = divmod(mt, per_clip)
        sp, t = divmod(r, t_len)
        return c * per_clip + t * s_ + sp

    for (hw, t_len, clips) in ((160 * 160, 8, 2), (320 * 320, 8, 2), (80 * 80, 5, 3), (40 * 40, 8, 2), (256, 3, 1)):
        n = clips * t_len * (-(-hw // 256))
        got = sorted(temporal_order(m, hw, t_len) for m in range(n))
        assert got == list(range(n)), (hw, t_len)
        if hw % 256 == 0:                      # neighbours in launch order = same pixels, consecutive frames
            a, b = temporal_order(0, hw, t_len), temporal_order(1, hw, t_len)
            assert b - a == hw // 256

    # persistent walk: workgroup wg of G handles wg, wg + G, ... < ntiles
    for ntiles, g in ((12800, 256), (3201, 256), (257, 256)):
        seen = sorted(t for wg in range(g) for t in range(wg, ntiles, g))
        assert seen == list(range(ntiles))


def test_cabi_rejects_bad_arguments_without_touching_the_gpu():
    """Error behaviour of the C ABI (include/uav_hip.h): argument checks return UAV_E* codes before any launch, so
    they can be exercised on a box without a GPU.  Pointers are dummies and are never dereferenced."""
    import ctypes as C
    from uav import _lib
    lib = _lib.load()
    EINVAL, EALIGN, ESHAPE = -1, -2, -3
    P = 0x1000                                   # non-null dummy "device pointer"
    base = dict(a1=P, c1=64, c2=0, w=P, out=P, out_stride=128, n_img=2, t_len=1, hi=8, wi=8, ho=8, wo=8, kt=1, kh=3, kw=3,
                stride=1, pad_h=1, pad_w=1, n=128, n_pad=128, k_pad=576, out_scale=1.0, zero_page=P)

    def conv(**kw):
        prm = _lib.ConvParams(**{**base, **kw})
        return lib.uav_conv_gemm_f16(C.byref(prm), None)
    assert conv(a1=None) == EINVAL and conv(zero_page=None) == EINVAL and conv(w=None) == EINVAL
    assert conv(c1=48) == ESHAPE                  # channels must be a multiple of 64 (or exactly 8)
    assert conv(c2=64) == EINVAL                  # second source announced but a2 missing
    assert conv(k_pad=512) == ESHAPE              # k_pad != taps * cin
    assert conv(n=126) == ESHAPE and conv(n=256) == ESHAPE       # n % 4, n > n_pad
    assert conv(out_stride=126) == EALIGN
    assert conv(upsample=1) == ESHAPE             # upsample needs ho = 2 hi
    assert conv(t_len=3) == ESHAPE                # n_img % t_len
    assert conv(flags=1, residual=P, res_stride=128) == ESHAPE   # GEGLU excludes residual
    assert conv(rowbias=P, rows_per_batch=0) == ESHAPE
    assert conv(flags=128) == EINVAL              # fp32-residual flag without a residual
    assert lib.uav_conv_gemm_f16(None, None) == EINVAL
    # fused GroupNorm statistics: offered only where every wave of the 256x256 kernel takes a fast epilogue
    big = dict(n_img=16, hi=64, wi=64, ho=64, wo=64, n=512, n_pad=512, out_stride=512, gn_groups=32)

    def gn_rows(**kw):
        return lib.uav_conv_gemm_gn_chunk_rows(C.byref(_lib.ConvParams(**{**base, **big, **kw})))
    assert gn_rows() == 64
    assert gn_rows(residual=P, res_stride=512, rowbias=P, rows_per_batch=4096, rowbias_stride=512) == 64
    assert gn_rows(flags=2, bias=P) == 64 and gn_rows(flags=2) == 0                 # fp32 out needs the bias fast path
    assert gn_rows(flags=2, bias=P, residual=P, res_stride=512) == 0                # fp32 out + fp16 residual: generic path
    assert gn_rows(flags=2 | 128, bias=P, residual=P, res_stride=512) == 64
    assert gn_rows(n=128, n_pad=256, out_stride=128) == 64 and gn_rows(n=64, n_pad=256, out_stride=64, gn_groups=16) == 0
    assert gn_rows(n_img=1, hi=8, wi=8, ho=8, wo=8) == 0                            # small grid: 128x128 kernel
    assert gn_rows(n_img=15, hi=63, ho=63, wi=63, wo=63) == 0                       # M % 64
    assert gn_rows(gn_groups=0) == 0 and gn_rows(gn_groups=48) == 0 and gn_rows(gn_groups=2) == 0    # 256 ch / group
    assert gn_rows(n=1536, n_pad=1536, out_stride=1536) == 0                        # 48 channels per group
    assert gn_rows(rowbias=P, rows_per_batch=100, rowbias_stride=512) == 0          # a wave tile would straddle batches
    assert gn_rows(flags=1, n=1024, n_pad=1024) == 0                                # GEGLU
    assert conv(**{**big, 'gn_partials': P, 'gn_groups': 48}) == ESHAPE                       # request that cannot be honoured
    assert gn_rows(out_map_w=64, out_map_sy=256, out_map_sx=2) == 0                 # strided output rows: no chunk runs
    assert conv(out_map_w=8, out_map_sy=32, out_map_sx=2, residual=P, res_stride=128) == ESHAPE   # row map excludes a residual
    assert conv(out_map_w=8, out_map_sy=32, out_map_sx=0) == ESHAPE and conv(out_map_w=-1) == ESHAPE
    assert lib.uav_groupnorm_finalize_partials(None, 8, 64, 64, 1, 512, 32, 1e-5, None, None, P, P, None) == EINVAL
    assert lib.uav_groupnorm_finalize_partials(P, 8, 64, 64, 1, 500, 32, 1e-5, None, None, P, P, None) == ESHAPE   # rows % 64
    assert lib.uav_groupnorm_finalize_partials(P, 9, 64, 64, 1, 512, 32, 1e-5, None, None, P, P, None) == ESHAPE   # chunk count
    assert lib.uav_cast_f32_f16(None, P, 8, None) == EINVAL and lib.uav_sft_fuse(P, P, None, P, 8, 1.0, 0, 0, None) == EINVAL
    # GroupNorm
    assert lib.uav_groupnorm_scale_shift(None, None, 0, 64, 0, 0, 64, 1, 16, 32, 1e-5, None, None, P, P, P, 1 << 20, None) == EINVAL
    assert lib.uav_groupnorm_scale_shift(P, None, 0, 60, 0, 0, 60, 1, 16, 30, 1e-5, None, None, P, P, P, 1 << 20, None) == ESHAPE
    assert lib.uav_groupnorm_scale_shift(P, None, 1, 64, 0, 0, 64, 1, 16, 7, 1e-5, None, None, P, P, P, 1 << 20, None) == ESHAPE
    assert lib.uav_groupnorm_scale_shift(P, None, 0, 64, 0, 0, 64, 1, 16, 32, 1e-5, None, None, P, P, P, 8, None) == EINVAL    # workspace too small
    assert lib.uav_groupnorm_scale_shift(P, P, 0, 64, 64, 7, 128, 2, 16, 32, 1e-5, None, None, P, P, P, 1 << 20, None) == ESHAPE   # x2_rows != half
    assert lib.uav_groupnorm_apply(P, P, 0, 64, 64, 7, 2, 16, P, P, 1, P, None, 0, None) == ESHAPE
    assert conv(c2=64, a2=P, k_pad=1152, a2_images=3) == ESHAPE                     # broadcast source: n_img must be 2 * a2_images
    # attention: null / shape / alignment
    assert lib.uav_attention_f16(None, 64, P, 64, P, 64, P, 64, 1, 8, 8, 1, 1, 64, 0.125, 0, P, None) == EINVAL
    assert lib.uav_attention_f16(P, 64, P, 64, P, 64, P, 64, 3, 8, 8, 2, 1, 64, 0.125, 0, P, None) == ESHAPE        # bq % q_per_kv
    assert lib.uav_attention_f16(P, 60, P, 64, P, 64, P, 64, 1, 8, 8, 1, 1, 64, 0.125, 0, P, None) == EALIGN
    assert lib.uav_attention_f16(P, 96, P, 96, P, 96, P, 96, 1, 8, 8, 1, 1, 96, 0.125, 0, P, None) == ESHAPE        # head_dim not built
    assert lib.uav_attention_f16(P, 64, P, 64, P, 64, P, 64, 1, 8, 16, 1, 1, 64, 0.125, 1, P, None) == ESHAPE       # causal needs lq == lk
    # temporal attention: T > 8 is not on the path
    assert lib.uav_temporal_attention_f16(P, P, 1, 9, 16, 512, 8, 0.125, P, P, 32, P, None) == ESHAPE
    assert lib.uav_resize_bilinear_f32(None, P, 1, 4, 4, 8, 8, 1.0, 1.0, None) == EINVAL
    assert lib.uav_version() > 0


def test_fused_groupnorm_partials_bookkeeping():
    """Host logic around the statistics a conv epilogue leaves on its output tensor (ops.GnPartials): they are honoured
    only for the very tensor object and version the conv returned, for the matching group count / width and a row count
    the 64-row chunks divide; duplicating the rows (CFG-shared head) duplicates them; a half-size second source is
    recognised as batch-broadcast and anything else is an error."""
    from uav import ops, _lib
    y = torch.zeros(256, 64, dtype=torch.float16)
    gn = ops.GnPartials(torch.arange(2 * 32 * 4, dtype=torch.float32).reshape(2, 32, 4), 64, 32, 64)
    ops._gn_attach(y, gn)
    assert ops._gn_partials_of(y, 32, 64, 128) is gn and ops._gn_partials_of(y, 32, 64, 256) is gn
    assert ops._gn_partials_of(y, 16, 64, 128) is None            # other group count
    assert ops._gn_partials_of(y, 32, 64, 96) is None             # instance not a whole number of chunks
    assert ops._gn_partials_of(y[:128], 32, 64, 128) is None      # a view is another object
    d = ops.duplicate_rows(y)
    gd = ops._gn_partials_of(d, 32, 64, 256)
    assert d.shape == (512, 64) and gd is not None and gd.ws.shape == (2, 32, 8)
    assert torch.equal(gd.ws[..., :4], gn.ws) and torch.equal(gd.ws[..., 4:], gn.ws)
    y.add_(1)                                                      # in-place update: stale
    assert ops._gn_partials_of(y, 32, 64, 128) is None
    assert getattr(ops.duplicate_rows(y), "_uav_gn", None) is None
    x = torch.zeros(8, 64, dtype=torch.float16)
    assert ops._gn_x2_rows(x, None, 8) == 0 and ops._gn_x2_rows(x, torch.zeros(8, 8), 8) == 0
    assert ops._gn_x2_rows(x, torch.zeros(4, 8), 8) == 4
    with pytest.raises(_lib.UavError):
        ops._gn_x2_rows(x, torch.zeros(3, 8), 8)


def test_upsample_phase_weights_identity():
    """"nearest 2x, then 3x3 conv (pad 1)" == four 2x2 sub-pixel phase convs (ops.upsample_phase_weights), exactly in exact
    arithmetic: checked in fp64 against F.interpolate + F.conv2d, borders included, and through the CPU stand-in of the
    kernel with its strided output rows (the call Upsample3D.run makes)."""
    import torch.nn.functional as F
    from uav import ops
    import cpu_ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 5, 7, generator=g, dtype=torch.float64)
    w = torch.randn(8, 64, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    ph = ops.upsample_phase_weights(w.float())
    out = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            xp = F.pad(x, (1 - px, px, 1 - py, py))                      # left/right, top/bottom: taps past the edge are zero
            out[:, :, py::2, px::2] = F.conv2d(xp, ph[py][px].double())
    assert (out - ref).abs().max().item() < 1e-5
    # the stand-in of the kernel call (fp16 rows in, fp32 rows out)
    rows = x.permute(0, 2, 3, 1).reshape(-1, 64).half()
    o = torch.zeros(2 * 10 * 14, 8)
    for py in range(2):
        for px in range(2):
            cw = ops.pack_conv(ph[py][px], None, n_store_align=4)
            cpu_ops.conv_gemm(rows, cw, n_img=2, t_len=1, hi=5, wi=7, pad=(0, 1 - py, 1 - px), out_hw=(5, 7), out_f32=True,
                              out=o, out_map=(7, 28, 2, py * 14 + px))
    ref16 = F.conv2d(F.interpolate(rows.float().reshape(2, 5, 7, 64).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"),
                     w.float(), padding=1)
    got = o.reshape(2, 10, 14, 8).permute(0, 3, 1, 2)
    assert ((got - ref16).norm() / ref16.norm()).item() < 2e-3             # fp16 rounding of the packed phase weights


def test_geglu_packing_order():
    from uav import ops
    f, k = 64, 64
    w = torch.arange(2 * f, dtype=torch.float32)[:, None].expand(2 * f, k).contiguous() / 1024
    b = torch.arange(2 * f, dtype=torch.float32)
    cw = ops.pack_conv(w, b, geglu=True, device="cpu")
    # packed rows: [32 value | 32 gate] per block of 32 features
    assert torch.equal(cw.bias[:32], b[:32]) and torch.equal(cw.bias[32:64], b[f:f + 32])
    assert torch.equal(cw.bias[64:96], b[32:64]) and torch.equal(cw.bias[96:128], b[f + 32:f + 64])


def test_window_schedule_and_scheduler_coefficients():
    from models_video.pipeline_upscale_a_video import window_schedule
    from models_video.scheduling_ddim import DDIMScheduler
    for t in (3, 8, 9, 10, 14, 27, 32, 47):
        assert window_schedule(t) == O.window_schedule(t)
    assert window_schedule(32)[-2:] == [(24, 32), (24, 32)]          # the reference's duplicate tail window
    g = torch.Generator().manual_seed(0)
    eps, x = torch.randn(64, generator=g), torch.randn(64, generator=g)
    for pred in ("v_prediction", "epsilon", "sample"):
        kw = dict(GC.SCHED, prediction_type=pred)
        sch = DDIMScheduler(**kw); sch.set_timesteps(30)
        ora = O.DDIM(**kw); ora.set_timesteps(30)
        for t in (958, 496, 1):
            cs, ce = sch.v0_coefficients(t)
            x0 = cs * x + ce * eps
            assert (x0 - ora.step_v0(eps, t, x)).abs().max().item() < 1e-5
            c0, cd, em, es, e0 = sch.vt_coefficients(t)
            prev = c0 * x0 + cd * (em * eps + es * x + e0 * x0)
            assert (prev - ora.step_vt(x0, eps, t, x)).abs().max().item() < 1e-5


def test_relative_position_bucket_table():
    from models_video.attention import RelativePositionBias
    rpb = RelativePositionBias(heads=8, max_distance=32)
    for n in (3, 8, 12):
        pos = torch.arange(n)
        ora = O.relative_position_bucket(pos[None, :] - pos[:, None], 32, 32)
        assert torch.equal(rpb.bucket_table(n), ora)


# ---------------------------------------------------------------------------------------------
# 4b. spatial tiling (the reference CLI's tile loop as a work list)
def test_tile_grid_matches_reference_cli_loop():
    """tests/golden/cli_tiles.json was produced by EXECUTING the reference's own tile loop
    (inference_upscale_a_video.py:207-303) around a recording pipeline (oracle/make_golden.py --tiles)."""
    from uav import tiling
    gold = json.load(open(os.path.join(GOLD, "cli_tiles.json")))
    assert len(gold) >= 8
    for key, ref in gold.items():
        hw, tile = key.split("_tile")
        h, w = map(int, hw.split("x"))
        mine = tiling.tile_grid(h, w, int(tile))
        assert [list(t.src) for t in mine] == [r["src"] for r in ref], key
        assert [list(t.dst) for t in mine] == [r["dst"] for r in ref], key
        assert [list(t.crop) for t in mine] == [r["crop"] for r in ref], key
        # the output boxes tile the 4x canvas exactly once
        cover = torch.zeros(4 * h, 4 * w, dtype=torch.int32)
        for t in mine:
            cover[t.dst[0]:t.dst[1], t.dst[2]:t.dst[3]] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
    assert tiling.needs_tiling(384, 384) and not tiling.needs_tiling(320, 320) and tiling.needs_tiling(64, 64, True)


def test_tile_grid_properties_random_sizes():
    """For any frame size the CLI would tile (and any tile size it accepts): output boxes partition the 4x canvas,
    every crop has its destination's shape and lies inside the 4x padded tile, every padded tile contains its core."""
    from hypothesis import given, settings, strategies as st
    from uav import tiling

    @settings(max_examples=200, deadline=None)
    @given(st.integers(130, 1300), st.integers(130, 2200), st.sampled_from([128, 192, 256, 320, 384]))
    def check(h, w, tile):
        tiles = tiling.tile_grid(h, w, tile)
        area = 0
        seen = set()
        for t in tiles:
            sy0, sy1, sx0, sx1 = t.src
            dy0, dy1, dx0, dx1 = t.dst
            cy0, cy1, cx0, cx1 = t.crop
            assert 0 <= sy0 < sy1 <= h and 0 <= sx0 < sx1 <= w
            assert (cy1 - cy0, cx1 - cx0) == (dy1 - dy0, dx1 - dx0)
            assert 0 <= cy0 and cy1 <= 4 * (sy1 - sy0) and 0 <= cx0 and cx1 <= 4 * (sx1 - sx0)
            assert dy0 == 4 * sy0 + cy0 and dx0 == 4 * sx0 + cx0          # the crop is the same pixels, not a shifted copy
            area += (dy1 - dy0) * (dx1 - dx0)
            assert (dy0, dx0) not in seen
            seen.add((dy0, dx0))
        assert area == 16 * h * w                                        # disjoint (distinct origins on a grid) and complete
    check()


def test_window_schedule_properties():
    from hypothesis import given, settings, strategies as st
    from models_video.pipeline_upscale_a_video import window_schedule

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 200))
    def check(t):
        wins = window_schedule(t)
        covered = set()
        for (s, e) in wins:
            assert 0 <= s < e <= t and (e - s == 8 or t < 8)
            covered.update(range(s, e))
        assert covered == set(range(t))
        if t > 8:
            assert all(b[0] - a[0] in (0, 6) or b[1] == t for a, b in zip(wins, wins[1:]))
    check()


def test_oracle_tiled_pipeline_vs_reference_cli_fixture(unet_sd):
    """tests/golden/pipe_tiled_t2_68x160.pt: the reference CLI tile loop (executed from the reference tree) around the
    reference pipeline, tiny seeded models; here the oracle replays it tile by tile with uav.tiling's boxes and the
    shared generator's draw order."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from uav import tiling
    vsd = synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY)).state_dict(), seed=4321)
    t, h, w, tile = 2, 68, 160, 64
    clip = synth.synth_clip(1, t, h, w, seed=33)
    gen = torch.Generator().manual_seed(10)
    dim = GC.UNET_TINY["cross_attention_dim"]
    pe = torch.cat([synth.synth_prompt_embeds("n", dim), synth.synth_prompt_embeds("p", dim)])
    out = torch.zeros(1, 3, t, 4 * h, 4 * w)
    for tl in tiling.tile_grid(h, w, tile):
        sub = clip[:, :, :, tl.src[0]:tl.src[1], tl.src[2]:tl.src[3]]
        lr_noise = torch.randn(sub.shape, generator=gen); lat0 = torch.randn((1, 4) + tuple(sub.shape[2:]), generator=gen)
        with torch.no_grad():
            img, _ = O.pipeline_call(unet_sd, GC.UNET_TINY, vsd, GC.VAE3D_TINY, sub, pe, num_inference_steps=2,
                                     guidance_scale=6.0, noise_level=120, lr_noise=lr_noise, latents=lat0,
                                     scheduler_kwargs=GC.SCHED)
        out[:, :, :, tl.dst[0]:tl.dst[1], tl.dst[2]:tl.dst[3]] = img[:, :, :, tl.crop[0]:tl.crop[1], tl.crop[2]:tl.crop[3]]
    gold = torch.load(os.path.join(GOLD, "pipe_tiled_t2_68x160.pt"))
    assert (out[..., ::4, ::4] - gold["sub4"].float()).abs().max().item() < 2e-3        # fixture stored in fp16
    assert (out[..., :, 240:272] - gold["seam"].float()).abs().max().item() < 2e-3      # across the tile seam at x = 256


def test_oracle_duplicate_tail_window_vs_reference_fixture(unet_sd):
    """T = 14 visits window [6,14) twice (pipeline :601-634); the second visit re-blends, it is not an identity.
    Fixture: the reference pipeline's latents (oracle/make_golden.py --pipe14)."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import window_schedule
    assert window_schedule(14) == [(0, 8), (6, 14), (6, 14)]
    vsd = synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY)).state_dict(), seed=4321)
    t, h, w = 14, 16, 16
    clip = synth.synth_clip(1, t, h, w, seed=14)
    gen = torch.Generator().manual_seed(10)
    lr_noise = torch.randn(clip.shape, generator=gen); lat0 = torch.randn((1, 4, t, h, w), generator=gen)
    dim = GC.UNET_TINY["cross_attention_dim"]
    pe = torch.cat([synth.synth_prompt_embeds("n", dim), synth.synth_prompt_embeds("p", dim)])
    with torch.no_grad():
        _, lat = O.pipeline_call(unet_sd, GC.UNET_TINY, vsd, GC.VAE3D_TINY, clip, pe, num_inference_steps=2, guidance_scale=6.0,
                                 noise_level=120, lr_noise=lr_noise, latents=lat0, scheduler_kwargs=GC.SCHED)
    gold = torch.load(os.path.join(GOLD, "pipe_t14_dup_tail.pt"))
    assert rel_l2(lat, gold["latents"]) < 1e-3          # fixture stored in fp16


# ---------------------------------------------------------------------------------------------
# 4c. the REAL VideoUpscalePipeline.__call__ host logic on CPU: HIP elementwise ops replaced by their torch equivalents
#     (same arithmetic as csrc/elementwise.hip), UNet / VAE replaced by wrappers around the oracle's forward passes.
#     This runs the window loop, blends, CFG / DDIM stepping, decode chunking and rank sharding of the product class.
def _install_cpu_ops(ns):
    """torch stand-ins for the HIP elementwise ops the pipeline / schedulers call (tests/cpu_ops.py)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    cpu_ops.install()
    ns["cpu_ops"] = cpu_ops


def _restore_ops(ns):
    if "cpu_ops" in ns:
        ns["cpu_ops"].restore()


def _cpu_pipeline(unet_sd, vsd, calls=None, vae_cfg=None, propagator=None):
    import types
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from uav.standin_text import StandInTextEncoder, StandInTokenizer

    class OracleUNet(torch.nn.Module):
        config = types.SimpleNamespace(in_channels=7)

        def forward(self, sample, t, low_res, encoder_hidden_states=None, class_labels=None, cfg_shared_input=False):
            if calls is not None:
                calls.append((tuple(sample.shape), int(class_labels.reshape(-1)[0])))
            with torch.no_grad():
                out = O.unet_forward(unet_sd, GC.UNET_TINY, sample.float(), int(t), low_res.float(),
                                     encoder_hidden_states.float(), class_labels)
            return types.SimpleNamespace(sample=out.half())

    vcfg = vae_cfg or GC.VAE3D_TINY

    class OracleVAE(torch.nn.Module):
        config = types.SimpleNamespace(scaling_factor=vcfg.get("scaling_factor", 0.08333), latent_channels=4, out_channels=3)

        def decode(self, z, img, w_lr=1.0):
            with torch.no_grad():
                return types.SimpleNamespace(sample=O.vae_decode(vsd, vcfg, z.float(), img, w_lr))
    tok = StandInTokenizer()
    dim = GC.UNET_TINY["cross_attention_dim"]
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, dim, dtype=torch.float32), tokenizer=tok,
                                low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED), vae=OracleVAE(),
                                unet=OracleUNet(), propagator=propagator)
    return pipe.to("cpu")


def _vae_sd():
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    return synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY)).state_dict(), seed=4321)


@pytest.mark.parametrize("t,denoise", [(14, None), (5, 77)])
def test_pipeline_host_logic_on_cpu_vs_oracle(unet_sd, t, denoise):
    """T = 14: sliding windows with the duplicate tail window; T = 5 with an explicit denoise_level: the single-window
    branch must still condition on noise_level (reference :638).  Product pipeline class vs oracle pipeline."""
    ns = {}
    _install_cpu_ops(ns)
    try:
        vsd = _vae_sd()
        calls = []
        pipe = _cpu_pipeline(unet_sd, vsd, calls)
        h, w = 16, 16
        clip = synth.synth_clip(1, t, h, w, seed=14)
        out, lat = pipe("p", image=clip, generator=torch.Generator().manual_seed(10), num_inference_steps=2, guidance_scale=6.0,
                        noise_level=120, denoise_level=denoise, negative_prompt="n", return_dict=False)
        gen = torch.Generator().manual_seed(10)
        lr_noise = torch.randn(clip.shape, generator=gen); lat0 = torch.randn((1, 4, t, h, w), generator=gen)
        dim = GC.UNET_TINY["cross_attention_dim"]
        pe = torch.cat([synth.synth_prompt_embeds("n", dim), synth.synth_prompt_embeds("p", dim)])
        with torch.no_grad():
            oimg, olat = O.pipeline_call(unet_sd, GC.UNET_TINY, vsd, GC.VAE3D_TINY, clip, pe, num_inference_steps=2,
                                         guidance_scale=6.0, noise_level=120, lr_noise=lr_noise, latents=lat0,
                                         scheduler_kwargs=GC.SCHED)
        assert out.shape == (1, 3, t, 4 * h, 4 * w) and out.dtype == torch.float32
        assert rel_l2(lat, olat) < 2e-2                 # fp16 latents between steps on the product side
        if t > 8:
            assert len(calls) == 2 * 2                  # 2 unique windows x 2 steps: the duplicate is not re-evaluated
            assert all(shape[2] == 8 for shape, _ in calls)
        else:
            assert [lvl for _, lvl in calls] == [120, 120]     # not 77
    finally:
        _restore_ops(ns)


def test_pipeline_host_logic_with_propagation_vs_reference_fixture(unet_sd):
    """BASELINE config-3 flow on the product pipeline class (CPU stand-ins): T = 10 (two windows), video VAE, latent
    propagation at DDIM step 1 through the propagator call protocol (pipeline :655-657) — against the fixture produced
    by the reference pipeline itself (tests/golden/pipe_t10_vaevideo_prop.pt)."""
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    case = GC.PIPE_CASES["pipe_t10_vaevideo_prop"]
    vsd = synth.synth_state_dict(AutoencoderKLVideo.from_config(dict(GC.VAEVIDEO_TINY)).state_dict(), seed=4321)
    seen = []

    class OraclePropagator(torch.nn.Module):
        def forward(self, x0, flows_f, flows_b, interpolation="nearest", mode="fuse", fuse_scale=0.5, alpha1=0.001, alpha2=0.05):
            assert flows_f.dtype == flows_b.dtype == x0.dtype       # reference :651: `flows_bi[k].to(latents)`
            seen.append((interpolation, mode, fuse_scale, alpha1, alpha2, tuple(flows_f.shape)))
            with torch.no_grad():
                return O.propagation(x0.float(), flows_f.float(), flows_b.float(), interpolation, fuse_scale, alpha1, alpha2).half()
    ns = {}
    _install_cpu_ops(ns)
    try:
        pipe = _cpu_pipeline(unet_sd, vsd, vae_cfg=GC.VAEVIDEO_TINY, propagator=OraclePropagator())
        image, flows = GC.pipeline_inputs(case)
        out, lat = pipe(case["prompt"], image=image, flows_bi=flows, generator=torch.Generator().manual_seed(10),
                        num_inference_steps=case["steps"], guidance_scale=case["guidance"], noise_level=case["noise_level"],
                        negative_prompt=case["negative"], propagation_steps=list(case["propagation_steps"]), return_dict=False)
        gold = torch.load(os.path.join(GOLD, "pipe_t10_vaevideo_prop.pt"))
        assert seen == [("nearest", "fuse", 0.5, 0.001, 0.05, tuple(flows[0].shape))]       # once, at step 1
        assert rel_l2(lat, gold["latents"]) < 2e-2
        unsat = gold["images"].float().abs() < 0.999
        assert rel_l2(out[unsat], gold["images"].float()[unsat]) < 4e-2
    finally:
        _restore_ops(ns)


def test_upscale_tiled_with_real_pipeline_on_cpu_vs_reference_fixture(unet_sd):
    """uav.tiling.upscale_tiled driving the product pipeline class (CPU stand-ins as above) against the fixture made by
    the reference CLI loop + reference pipeline: tile boxes, shared-generator draw order and stitching, end to end."""
    from uav import tiling
    ns = {}
    _install_cpu_ops(ns)
    try:
        pipe = _cpu_pipeline(unet_sd, _vae_sd())
        clip = synth.synth_clip(1, 2, 68, 160, seed=33)
        out = tiling.upscale_tiled(pipe, "p", clip, None, torch.Generator().manual_seed(10), tile_size=64,
                                   num_inference_steps=2, guidance_scale=6.0, noise_level=120, negative_prompt="n")
        gold = torch.load(os.path.join(GOLD, "pipe_tiled_t2_68x160.pt"))
        for mine, ref in ((out[..., ::4, ::4], gold["sub4"].float()), (out[..., :, 240:272], gold["seam"].float())):
            unsat = ref.abs() < 0.999
            assert rel_l2(mine[unsat], ref[unsat]) < 3e-2          # fp16 latents between DDIM steps on the product side
    finally:
        _restore_ops(ns)


class _FakeTilePipeline:
    """CPU stand-in with the pipeline's draw order: LR noise, then latents, from the shared generator."""

    class _Cfg:
        latent_channels = 4

    def __init__(self):
        self.vae = type("V", (), {"config": self._Cfg})()
        self.text_encoder = type("T", (), {"dtype": torch.float32})()

    def __call__(self, prompt, image=None, flows_bi=None, generator=None, **kw):
        from models_video.pipeline_upscale_a_video import randn_tensor
        n = randn_tensor(image.shape, generator=generator, device=image.device, dtype=torch.float32)
        lat = randn_tensor((1, 4) + tuple(image.shape[2:]), generator=generator, device=image.device, dtype=torch.float32)
        up = torch.nn.functional.interpolate((image + 0.1 * n)[0].permute(1, 0, 2, 3), scale_factor=4, mode="nearest")
        up = up.permute(1, 0, 2, 3)[None] + torch.nn.functional.interpolate(lat[0, :1].permute(1, 0, 2, 3), scale_factor=4).permute(1, 0, 2, 3)[None]
        return type("O", (), {"images": up})()


def _serial_cli_loop(pipe, vframes, gen, tile):
    """The CLI loop itself, restated for the test: one shared generator, tiles in order."""
    from uav import tiling
    out = vframes.new_zeros(vframes.shape[:3] + (vframes.shape[3] * 4, vframes.shape[4] * 4))
    for t in tiling.tile_grid(vframes.shape[3], vframes.shape[4], tile):
        res = pipe("p", image=vframes[:, :, :, t.src[0]:t.src[1], t.src[2]:t.src[3]], generator=gen).images
        out[:, :, :, t.dst[0]:t.dst[1], t.dst[2]:t.dst[3]] = res[:, :, :, t.crop[0]:t.crop[1], t.crop[2]:t.crop[3]]
    return out


def test_upscale_tiled_serial_equals_cli_loop():
    from uav import tiling
    g = torch.Generator().manual_seed(4)
    vframes = torch.randn(1, 3, 2, 96, 150, generator=g)
    pipe = _FakeTilePipeline()
    ref = _serial_cli_loop(pipe, vframes, torch.Generator().manual_seed(10), 64)
    gen = torch.Generator().manual_seed(10)
    out = tiling.upscale_tiled(pipe, "p", vframes, None, gen, tile_size=64)
    assert torch.equal(out, ref)
    # replaying the draws reproduces the generator state each tile starts from
    gen2 = torch.Generator().manual_seed(10)
    tiles = tiling.tile_grid(96, 150, 64)
    states = tiling.generator_states(gen2, tiles, 2, 3, 4, torch.float32, "cpu")
    assert len(states) == len(tiles) and torch.equal(gen2.get_state(), gen.get_state())


# ---------------------------------------------------------------------------------------------
# 5. multi-GPU helpers on gloo, world_size 2
def _dist_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
    torch.set_num_threads(max(1, min(4, (os.cpu_count() or 2) // 2)))     # two workers share the host
    from uav import dist as D
    w, r, _ = D.init("gloo")
    clips = list(range(7))
    mine = D.shard(clips, r, w)
    D.barrier()
    elapsed = D.max_over_ranks(1.0 + r)
    total = D.sum_over_ranks(len(mine))
    gathered = D.gather_to_rank0(mine)
    # --- one long clip over the ranks (BASELINE config 4): windows of a DDIM step + decode chunks via sharded_map ---
    from models_video.pipeline_upscale_a_video import window_schedule
    t_total = 32
    wins = window_schedule(t_total)
    uniq = [w_ for k, w_ in enumerate(wins) if w_ not in wins[:k]]
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(2, 4, t_total, 6, 5, generator=g)
    calls = []

    def fake_unet(se):                       # stands in for UNetVideoModel on one 8-frame window (CPU)
        calls.append(se)
        x = lat[:, :, se[0]:se[1]]
        return torch.tanh(x * 1.7 + x.mean(dim=2, keepdim=True)) * (1 + 0.01 * se[0])

    def blend(outs):                         # the running 0.5/0.5 overlap blend of pipeline_upscale_a_video.py
        eps, written = torch.zeros_like(lat), [False] * t_total
        for (s_, e_) in wins:
            o = outs[(s_, e_)]
            for k, idx in enumerate(range(s_, e_)):
                eps[:, :, idx] = o[:, :, k] if not written[idx] else 0.5 * eps[:, :, idx] + 0.5 * o[:, :, k]
                written[idx] = True
        return eps
    sharded = blend(dict(zip(uniq, D.sharded_map(uniq, fake_unet))))
    n_local = len(calls)
    serial = blend({w_: fake_unet(w_) for w_ in uniq})
    chunks = D.sharded_map(list(range(0, t_total, 3)), lambda s_: torch.full((1, 3, 3, 2, 2), float(s_)))
    single = D.sharded_map([5], lambda s_: torch.full((2,), float(s_)))          # one item: no collective
    # --- spatial tiles of one clip over the ranks (BASELINE config 5): bit-identical to the serial CLI loop ---
    from uav import tiling
    gv = torch.Generator().manual_seed(4)
    vframes = torch.randn(1, 3, 2, 96, 150, generator=gv)
    pipe = _FakeTilePipeline()
    tiled = tiling.upscale_tiled(pipe, "p", vframes, None, torch.Generator().manual_seed(10), tile_size=64)
    tiles_same = torch.equal(tiled, _serial_cli_loop(pipe, vframes, torch.Generator().manual_seed(10), 64))
    # --- the REAL pipeline class with shard_windows over the two ranks (CPU ops, oracle-backed UNet / VAE) ---
    from models_video.unet_video import UNetVideoModel
    usd = synth.synth_state_dict(UNetVideoModel.from_config(dict(GC.UNET_TINY)).state_dict(), seed=1234)
    vsd = _vae_sd()
    ns = {}
    _install_cpu_ops(ns)
    ucalls = []
    rp = _cpu_pipeline(usd, vsd, ucalls)
    clip14 = synth.synth_clip(1, 14, 16, 16, seed=14)
    kw14 = dict(image=clip14, num_inference_steps=2, guidance_scale=6.0, noise_level=120, negative_prompt="n", return_dict=False)
    img_serial, lat_serial = rp("p", generator=torch.Generator().manual_seed(10), **kw14)
    n_serial = len(ucalls)
    rp.shard_windows = True
    img_shard, lat_shard = rp("p", generator=torch.Generator().manual_seed(10), **kw14)
    pipe_same = torch.equal(img_serial, img_shard) and torch.equal(lat_serial, lat_shard)
    n_shard = len(ucalls) - n_serial
    # --- (window x guidance branch) units: 2 windows x 2 branches x 2 steps = 8 batch-1 UNet calls, 4 per rank; bit-identical
    # to the SAME decomposition evaluated serially, and equal to the batch-2 schedule up to fp32 summation order ---
    rp.shard_cfg = True
    n0 = len(ucalls)
    img_cfg, lat_cfg = rp("p", generator=torch.Generator().manual_seed(10), **kw14)
    n_cfg = len(ucalls) - n0
    cfg_shapes = sorted({c[0][0] for c in ucalls[n0:]})
    import models_video.pipeline_upscale_a_video as PM
    real_map = PM.D.sharded_map
    PM.D.sharded_map = lambda items, fn, **kw: [fn(it) for it in items]
    try:
        img_cfg_serial, lat_cfg_serial = rp("p", generator=torch.Generator().manual_seed(10), **kw14)
    finally:
        PM.D.sharded_map = real_map
    cfg_same = torch.equal(img_cfg, img_cfg_serial) and torch.equal(lat_cfg, lat_cfg_serial)
    cfg_close = float((lat_cfg.float() - lat_serial.float()).norm() / lat_serial.float().norm())
    # one 8-frame clip (single window): the two guidance branches are the two units
    clip8 = synth.synth_clip(1, 8, 16, 16, seed=8)
    kw8 = dict(kw14, image=clip8)
    n0 = len(ucalls)
    img8, lat8 = rp("p", generator=torch.Generator().manual_seed(10), **kw8)
    n8 = len(ucalls) - n0
    rp.shard_windows = rp.shard_cfg = False
    img8_ref, lat8_ref = rp("p", generator=torch.Generator().manual_seed(10), **kw8)
    close8 = float((lat8.float() - lat8_ref.float()).norm() / lat8_ref.float().norm())
    _restore_ops(ns)
    q.put((r, mine, elapsed, total, gathered, torch.equal(sharded, serial), n_local, len(uniq),
           [float(c.flatten()[0]) for c in chunks], float(single[0][0]), tiles_same, pipe_same, n_serial, n_shard,
           cfg_same, n_cfg, cfg_shapes, cfg_close, n8, close8))
    D.finalize()


def test_dist_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, e0, t0, g0, *x0), (r1, m1, e1, t1, g1, *x1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]
    assert e0 == e1 == 2.0 and t0 == t1 == 7.0
    assert g0 == [[0, 2, 4, 6], [1, 3, 5]] and g1 is None
    # window-sharded long clip: bit-identical to the serial schedule on BOTH ranks, each rank ran only its share
    for x in (x0, x1):
        same, n_local, n_uniq, chunk_ids, single, tiles_same, pipe_same, n_serial, n_shard, cfg_same, n_cfg, cfg_shapes, cfg_close, n8, close8 = x
        assert cfg_same                            # (window x guidance branch) units: sharded == serial, bit for bit
        assert n_cfg == 4 and cfg_shapes == [1]    # 8 batch-1 UNet calls per clip, 4 on each rank
        assert cfg_close < 1e-3 and close8 < 1e-3  # == the batch-2 schedule up to summation order (fp16 latents between steps)
        assert n8 == 2                             # ONE 8-frame clip on 2 ranks: one guidance branch per rank and step
        assert tiles_same                          # tile-sharded clip == serial CLI loop, on both ranks
        assert pipe_same                           # VideoUpscalePipeline(shard_windows=True) == serial call, bit for bit
        assert n_serial == 4 and n_shard == 2      # 2 unique windows x 2 steps; each rank evaluates one window per step
        assert same and n_uniq == 5 and chunk_ids == [float(s) for s in range(0, 32, 3)] and single == 5.0
    assert x0[1] == 3 and x1[1] == 2           # 5 unique windows dealt 3 / 2


def test_bench_predicted_scaling_on_the_line():
    """`config.predicted_scaling`: clip-parallel = N x the per-GPU rate at the assumed efficiency; sharded
    long clip = ceil(units / N) window units + ceil(chunks / N) decode chunks per GPU, 5 (or 10 with --shard-cfg) units for 32 frames."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    a = types.SimpleNamespace(shard_windows=False, shard_cfg=False, frames=8, ddim_steps=30)
    p = bench.predicted_scaling(a, 1, 1.0)
    assert p["frames_per_s"]["1"] == 1.0 and abs(p["frames_per_s"]["8"] - 8 * 0.97) < 1e-9
    a = types.SimpleNamespace(shard_windows=True, shard_cfg=False, frames=32, ddim_steps=30, height=320, width=320)
    p = bench.predicted_scaling(a, 1, 0.7)
    assert p["units_per_step"] == 5 and p["decode_chunks"] == 11 and "approximate" not in p["unit_costs"]
    a.height = 256
    assert "approximate" in bench.predicted_scaling(a, 1, 0.7)["unit_costs"]      # the unit costs are those of the default shape
    a.height = 320
    assert 1.6 < p["speedup_vs_1_gpu"]["2"] < 1.7 and 2.4 < p["speedup_vs_1_gpu"]["4"] < 2.6 and 4.9 < p["speedup_vs_1_gpu"]["8"] < 5.1
    a.shard_cfg = True
    p = bench.predicted_scaling(a, 1, 0.7)
    assert p["units_per_step"] == 10 and 3.2 < p["speedup_vs_1_gpu"]["4"] < 3.5


def test_publish_acquire_are_noops_without_a_gpu():
    """uav.engine.publish / acquire (event-ordered shared objects) must not touch the HIP runtime on a CPU-only host."""
    from uav import engine as E
    obj = object()
    assert E.publish(obj) is obj and E.acquire(obj) is obj and not E._PENDING


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="needs hipcc + llvm-readelf (ROCm)")
def test_asm_scheduled_conv_kernels_have_no_scratch():
    """the rotated k-step of conv_gemm256i_kernel<6,*> and the four-wave conv_gemm256w_kernel keep DMA targets and
    operand fragments live across inline-asm statements; a compiler spill between them would read stale data silently.  The
    build audit (uav/build.py audit_conv_scratch, run on every build) reads the code-object metadata of the library: no private
    segment, no spilled VGPR in either kernel family; the accumulators of the four-wave kernel are the full 256-entry AGPR file."""
    from uav import build
    build.build(verbose=False)
    build.audit_conv_scratch()
    meta = build.kernel_metadata()
    w4 = {k: v for k, v in meta.items() if "conv_gemm256w_kernel" in k}
    assert len(w4) >= 4
    for name, md in w4.items():
        assert md["agpr_count"] == "256" and md["max_flat_workgroup_size"] == "256", (name, md)


def test_magic_division_of_the_four_wave_conv_setup():
    """conv_gemm.hip conv_magic: n / d == umulhi(n, mul) >> sh for every n < 2^31 (the output-pixel index range the four-wave
    kernel's setup divides by Ho*Wo, Wo and T).  Python restatement of the host function, checked on the divisors the models use,
    on random divisors, and on the values around every multiple boundary where a wrong multiplier fails first."""
    import random

    def magic(d):
        if d <= 1:
            return 0, 32
        s = 0
        while (1 << s) < d:
            s += 1
        return ((1 << (31 + s)) // d + 1) & 0xFFFFFFFFFFFF, s - 1

    def div(n, mul, sh):
        return n if sh == 32 else ((n * mul) >> 32) >> sh

    rng = random.Random(7)
    ds = [2, 3, 5, 7, 8, 14, 40, 80, 160, 320, 640, 1280, 40 * 40, 80 * 80, 160 * 160, 320 * 320, 1280 * 1280, 2560 * 2560, 65535, 65535 * 65535 // 3]
    ds += [rng.randrange(2, 1 << 31) for _ in range(200)] + [(1 << k) + e for k in range(1, 31) for e in (-1, 0, 1) if (1 << k) + e >= 2]
    for d in ds:
        mul, sh = magic(d)
        assert mul < (1 << 32), d
        ns = [0, 1, d - 1, d, d + 1, (1 << 31) - 1, ((1 << 31) - 1) // d * d, ((1 << 31) - 1) // d * d - 1]
        ns += [rng.randrange(0, 1 << 31) for _ in range(50)] + [q * d + e for q in (rng.randrange(0, (1 << 31) // d + 1) for _ in range(20)) for e in (-1, 0)]
        for n in ns:
            if 0 <= n < (1 << 31):
                assert div(n, mul, sh) == n // d, (n, d)
