"""RAFT (K11) parity on the GPU: fp32 kernels against PyTorch fp32 references of the same ops, and the
drop-in RAFT / RAFT_bi against the oracle restatement (which matches the reference's RAFT module
bit-for-bit on CPU, tests/golden/PINNING.json) and the reference-generated fixture."""
import json
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.fixture(scope="module")
def ops(dev):
    from uav import ops as _ops
    return _ops


@pytest.mark.parametrize("cin,cout,k,stride,pad,act", [
    (3, 64, (7, 7), 2, (3, 3), 1), (64, 96, (3, 3), 2, (1, 1), 0), (96, 96, (3, 3), 1, (1, 1), 1),
    (128, 256, (1, 1), 1, (0, 0), 0), (2, 128, (7, 7), 1, (3, 3), 1), (256, 2, (3, 3), 1, (1, 1), 0),
    (64, 96, (1, 1), 2, (0, 0), 0),
])
def test_conv_f32(ops, dev, cin, cout, k, stride, pad, act):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    n, h, w = 3, 22, 18
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, *k, generator=g) * (cin * k[0] * k[1]) ** -0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    ref = F.conv2d(x, wt, b, stride, pad)
    if act == 1:
        ref = F.relu(ref)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin)
    cw = ops.pack_conv_f32(wt, b, device=dev)
    if cw.cin_p != cin:
        rows = F.pad(rows, (0, cw.cin_p - cin))
    y = ops.conv_gemm_f32(rows.contiguous(), cw, n_img=n, hi=h, wi=w, stride=stride, pad=pad, act=act)
    ho, wo = ref.shape[-2:]
    y4 = y[:, :cout].reshape(n, ho, wo, cout).permute(0, 3, 1, 2)
    assert rel_l2(y4, ref) < 2e-5


def test_conv_f32_two_sources_gates_and_view_output(ops, dev):
    g = torch.Generator().manual_seed(3)
    n, h, w = 2, 16, 20
    a = torch.randn(n, 128, h, w, generator=g).to(dev); b2 = torch.randn(n, 256, h, w, generator=g).to(dev)
    wz = (torch.randn(256, 384, 1, 5, generator=g) * 0.02).to(dev); bz = torch.randn(256, generator=g).to(dev)
    ref = torch.sigmoid(F.conv2d(torch.cat([a, b2], 1), wz, bz, padding=(0, 2)))
    ra = a.permute(0, 2, 3, 1).reshape(-1, 128).contiguous(); rb = b2.permute(0, 2, 3, 1).reshape(-1, 256).contiguous()
    y = ops.conv_gemm_f32(ra, ops.pack_conv_f32(wz, bz, device=dev), a2=rb, n_img=n, hi=h, wi=w, pad=(0, 2), act=2)
    assert rel_l2(y.reshape(n, h, w, 256).permute(0, 3, 1, 2), ref) < 2e-5
    # 126-channel conv written into columns 128.. of a 256-wide row buffer, tanh epilogue, 324 -> 352 padded input
    w2 = (torch.randn(126, 128, 3, 3, generator=g) * 0.03).to(dev); bb = torch.randn(126, generator=g).to(dev)
    ref2 = torch.tanh(F.conv2d(a, w2, bb, padding=1)).permute(0, 2, 3, 1).reshape(-1, 126)
    X = torch.full((n * h * w, 256), 7.0, device=dev)
    ops.conv_gemm_f32(ra, ops.pack_conv_f32(w2, bb, device=dev), n_img=n, hi=h, wi=w, act=4, out=X[:, 128:256])
    assert rel_l2(X[:, 128:254], ref2) < 2e-5 and float(X[:, :128].min()) == 7.0 and float(X[:, 254:].abs().max()) == 0.0
    c324 = torch.randn(n * h * w, 324, generator=g).to(dev)
    w3 = (torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(dev)
    y3 = ops.conv_gemm_f32(F.pad(c324, (0, 28)).contiguous(), ops.pack_conv_f32(w3, None, device=dev, cin_pad_to=352), n_img=n, hi=h, wi=w)
    assert rel_l2(y3, c324 @ w3[:, :, 0, 0].t()) < 2e-5


def test_instnorm_pool_gates(ops, dev):
    g = torch.Generator().manual_seed(4)
    n, hw, c = 3, 300, 96
    x = (torch.randn(n, c, hw, generator=g) * 2 + 1).to(dev)
    ref = F.relu(F.instance_norm(x, eps=1e-5)).permute(0, 2, 1).reshape(-1, c)
    y = ops.instnorm_f32(x.permute(0, 2, 1).reshape(-1, c).contiguous(), n_img=n, hw=hw, relu=True)
    assert rel_l2(y, ref) < 1e-5
    vol = torch.randn(50, 12, 10, generator=g).to(dev)
    p = ops.avgpool2_f32(vol.reshape(50, -1).contiguous(), 120, 12, 10, 50)
    assert rel_l2(p, F.avg_pool2d(vol[:, None], 2, 2).reshape(50, -1)) < 1e-6
    zr = torch.rand(40, 256, generator=g).to(dev); h = torch.randn(40, 128, generator=g).to(dev); q = torch.randn(40, 128, generator=g).to(dev)
    assert rel_l2(ops.gru_rh_f32(zr, h), zr[:, 128:] * h) < 1e-6
    href = (1 - zr[:, :128]) * h + zr[:, :128] * q
    assert rel_l2(ops.gru_blend_f32(zr, q, h.clone()), href) < 1e-6


def test_corr_lookup_and_convex_upsample(ops, dev):
    import uav_oracle as O
    g = torch.Generator().manual_seed(6)
    n, h, w = 2, 16, 20
    f1 = torch.randn(n, 256, h, w, generator=g); f2 = torch.randn(n, 256, h, w, generator=g)
    pyr = O.raft_corr_pyramid(f1, f2)
    coords = torch.stack(torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")[::-1], 0).float()[None].repeat(n, 1, 1, 1)
    coords = coords + 3 * torch.randn(coords.shape, generator=g)
    ref = O.raft_corr_lookup(pyr, coords).permute(0, 2, 3, 1).reshape(-1, 324)
    levels = [p_.reshape(n * h * w, -1).contiguous().to(dev) for p_ in pyr]
    hs, ws = [p_.shape[-2] for p_ in pyr], [p_.shape[-1] for p_ in pyr]
    crow = torch.zeros(n * h * w, 4); crow[:, :2] = coords.permute(0, 2, 3, 1).reshape(-1, 2)
    out = torch.zeros(n * h * w, 352, device=dev)
    ops.corr_lookup_f32(levels, [l.shape[1] for l in levels], hs, ws, crow.to(dev), out)
    assert rel_l2(out[:, :324], ref) < 1e-5 and float(out[:, 324:].abs().max()) == 0.0
    flow = torch.randn(n, 2, h, w, generator=g); mask = torch.randn(n, 576, h, w, generator=g)
    ref_up = O.raft_upsample_flow(flow, mask)
    frow = torch.zeros(n * h * w, 4); frow[:, :2] = flow.permute(0, 2, 3, 1).reshape(-1, 2)
    up = ops.convex_upsample_f32(frow.to(dev), mask.permute(0, 2, 3, 1).reshape(-1, 576).contiguous().to(dev), n, h, w)
    assert rel_l2(up, ref_up) < 1e-5


def test_raft_bi_vs_oracle_and_reference_fixture(dev):
    import synth
    import uav_oracle as O
    from models_video.RAFT.raft_bi import RAFT_bi
    rb = RAFT_bi(model_path=None, device="cpu")
    sd = synth.synth_state_dict(rb.fix_raft.state_dict(), seed=777)
    rb.fix_raft.load_state_dict(sd)
    rb = rb.to(dev)
    clip = synth.synth_clip(1, 3, 128, 160, seed=5, motion=(2, 1))
    ff, fb = rb(clip.to(dev), iters=4)
    with torch.no_grad():
        off, ofb = O.raft_bi_forward(sd, clip, iters=4)
    assert ff.shape == (1, 2, 2, 128, 160) and fb.shape == ff.shape
    e_f, e_b = rel_l2(ff, off), rel_l2(fb, ofb)
    # fp32 on both sides (exact-fp32 MFMA), recurrent 4-iteration GRU: measured 3.0e-7 / 2.9e-7 (rounds 3-4); the bar is 1e-5
    # (the old 2e-3 would have passed a regression of three orders of magnitude)
    assert e_f < 1e-5 and e_b < 1e-5, (e_f, e_b)
    gold = torch.load(os.path.join(GOLD, "raft_bi_t3_128x160.pt"))          # reference outputs STORED as fp16: 2^-11 / sqrt(3) = 2.8e-4 of rounding
    assert rel_l2(ff, gold["forward"]) < 4e-4 and rel_l2(fb, gold["backward"]) < 4e-4
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(case="raft_bi_t3_128x160_iters4", rel_l2_fwd_vs_oracle=e_f, rel_l2_bwd_vs_oracle=e_b)) + "\n")


def test_resize_bilinear_matches_interpolate(dev):
    from uav import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 4, 45, 52, generator=g)
    ref = F.interpolate(x.reshape(2, 12, 45, 52), (48, 56), mode="bilinear")
    out = ops.resize_bilinear_f32(x.to(dev), 48, 56)
    assert out.shape == (2, 3, 4, 48, 56)
    assert (out.cpu().reshape(2, 12, 48, 56) - ref).abs().max().item() < 1e-5       # fp32 both sides
    flow = torch.randn(4, 2, 48, 56, generator=g) * 5
    ref = F.interpolate(flow, (45, 52), mode="bilinear")
    ref[:, :, 0] *= 45 / 48; ref[:, :, 1] *= 52 / 56                                # the reference's row-indexed rescale
    out = ops.resize_bilinear_f32(flow.to(dev), 45, 52, row0_scale=45 / 48, row1_scale=52 / 56)
    assert (out.cpu() - ref).abs().max().item() < 1e-4


def test_raft_bi_non_multiple_of_8_vs_oracle_and_reference_fixture(dev):
    """H, W not multiples of 8: pre-resize + flow resize path (reference raft_bi.py:49-53,11-16,62-63)."""
    import synth
    import uav_oracle as O
    from models_video.RAFT.raft_bi import RAFT_bi
    rb = RAFT_bi(model_path=None, device="cpu")
    sd = synth.synth_state_dict(rb.fix_raft.state_dict(), seed=777)
    rb.fix_raft.load_state_dict(sd)
    rb = rb.to(dev)
    clip = synth.synth_clip(1, 3, 132, 164, seed=5, motion=(2, 1))
    ff, fb = rb(clip.to(dev), iters=3)
    with torch.no_grad():
        off, ofb = O.raft_bi_forward(sd, clip, iters=3)
    assert ff.shape == (1, 2, 2, 132, 164) and fb.shape == ff.shape
    e_f, e_b = rel_l2(ff, off), rel_l2(fb, ofb)
    assert e_f < 1e-5 and e_b < 1e-5, (e_f, e_b)          # measured 3.0e-7 / 3.1e-7
    gold = torch.load(os.path.join(GOLD, "raft_bi_t3_132x164.pt"))          # fp16-stored fixture (see above)
    assert rel_l2(ff, gold["forward"]) < 4e-4 and rel_l2(fb, gold["backward"]) < 4e-4
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(case="raft_bi_t3_132x164_iters3", rel_l2_fwd_vs_oracle=e_f, rel_l2_bwd_vs_oracle=e_b)) + "\n")


def test_pipeline_with_raft_flows_and_propagation(dev):
    """BASELINE config-3 style run at reduced width: RAFT_bi flows -> x0-space propagation at one
    DDIM step, engine vs oracle (both consume their OWN RAFT flows)."""
    import golden_cases as GC
    import synth
    import uav_oracle as O
    from models_video.RAFT.raft_bi import RAFT_bi
    from models_video.autoencoder_kl_cond_video import AutoencoderKLVideo
    from models_video.pipeline_upscale_a_video import VideoUpscalePipeline
    from models_video.propagation_module import Propagation
    from models_video.scheduling_ddim import DDIMScheduler, DDPMScheduler
    from models_video.unet_video import UNetVideoModel
    from uav.standin_text import StandInTextEncoder, StandInTokenizer
    t, h, w = 3, 128, 128
    clip = synth.synth_clip(1, t, h, w, seed=9, motion=(2, 1))
    rb = RAFT_bi(model_path=None, device="cpu")
    rsd = synth.synth_state_dict(rb.fix_raft.state_dict(), seed=777)
    # damp the random flow head so that flows stay in a few-pixel range (nearest warp far from clamping)
    for k in ("update_block.flow_head.conv2.weight", "update_block.flow_head.conv2.bias"):
        rsd[k] = rsd[k] * 0.05
    rb.fix_raft.load_state_dict(rsd); rb = rb.to(dev)
    flows = rb.forward_slicing(clip.to(dev), iters=3)
    with torch.no_grad():
        oflows = O.raft_bi_forward(rsd, clip, iters=3)
    assert rel_l2(flows[0], oflows[0]) < 1e-3
    unet = UNetVideoModel.from_config(dict(GC.UNET_TINY)); usd = synth.synth_state_dict(unet.state_dict(), seed=1234)
    unet.load_state_dict(usd); unet = unet.to(dev).eval()
    vae = AutoencoderKLVideo.from_config(dict(GC.VAE3D_TINY)); vsd = synth.synth_state_dict(vae.state_dict(), seed=4321)
    vae.load_state_dict(vsd); vae = vae.to(dev).eval()
    tok = StandInTokenizer()
    prop = Propagation(4, learnable=False); prop.coord_f16 = False
    dim = GC.UNET_TINY["cross_attention_dim"]
    pipe = VideoUpscalePipeline(text_encoder=StandInTextEncoder(tok, dim, dtype=torch.float32), tokenizer=tok,
                                low_res_scheduler=DDPMScheduler(), scheduler=DDIMScheduler(**GC.SCHED), vae=vae, unet=unet,
                                propagator=prop).to(dev)
    gen = torch.Generator().manual_seed(10)
    out, lat = pipe("a prompt", image=clip.to(dev), flows_bi=list(flows), generator=gen, num_inference_steps=2, guidance_scale=6.0,
                    noise_level=120, negative_prompt="bad", propagation_steps=[1], return_dict=False)
    gen = torch.Generator().manual_seed(10)
    lr_noise = torch.randn(clip.shape, generator=gen); lat0 = torch.randn((1, 4, t, h, w), generator=gen)
    pe = torch.cat([synth.synth_prompt_embeds("bad", dim), synth.synth_prompt_embeds("a prompt", dim)])
    with torch.no_grad():
        _, olat = O.pipeline_call(usd, GC.UNET_TINY, vsd, GC.VAE3D_TINY, clip, pe, num_inference_steps=2, guidance_scale=6.0,
                                  noise_level=120, lr_noise=lr_noise, latents=lat0, flows_bi=list(oflows), propagation_steps=(1,),
                                  scheduler_kwargs=GC.SCHED, decode=False)
    # nearest-neighbour warps may pick a different source pixel where the two flow fields differ in the
    # last ulp around .5: compare robustly (fraction of latent elements off by more than 5e-2)
    bad = ((lat.float().cpu() - olat).abs() > 5e-2).float().mean().item()
    assert bad < 2e-2, f"{bad} of latent elements differ"
    assert out.shape == (1, 3, t, 4 * h, 4 * w)
