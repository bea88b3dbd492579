"""Per-kernel parity tests (GPU): every HIP kernel of libuav_hip.so, called through the C ABI,
against a plain PyTorch fp32 reference of the same op on identical fp16-representable inputs.

Tolerances (stated per test): fp16 storage of outputs carries 2^-11 relative rounding, MFMA
accumulates in fp32 -> rel-L2 <= 2e-3 for fp16 outputs, <= 1e-4 for fp32 outputs.
"""
import os
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.float(); b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def to_rows(x):
    """(N,C,H,W) -> channels-last rows [N*H*W][C] fp16."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous().half()


def from_rows(y, n, h, w):
    return y.reshape(n, h, w, -1).permute(0, 3, 1, 2).float()


def h16(*shape, dev, scale=1.0, gen=None):
    return (torch.randn(*shape, generator=gen, device="cpu") * scale).half().float().to(dev)


@pytest.fixture(scope="module")
def ops(dev):
    from uav import ops as _ops, _lib
    lib = _lib.load()
    assert lib.uav_device_check(0, None) == 0, "libuav_hip.so must run on gfx950"
    return _ops


# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # name, cin, cout, (kt,kh,kw), stride, upsample, n_img, t_len, h, w
    ("3x3_c64", 64, 128, (1, 3, 3), 1, False, 4, 2, 24, 20),
    ("3x3_c256_n256_tail", 256, 256, (1, 3, 3), 1, False, 2, 2, 17, 19),
    ("3x3_stride2", 128, 128, (1, 3, 3), 2, False, 2, 1, 16, 24),
    ("3x3_upsample", 64, 64, (1, 3, 3), 1, True, 2, 2, 9, 11),
    ("1x1", 192, 128, (1, 1, 1), 1, False, 2, 1, 13, 7),
    ("t3", 64, 64, (3, 1, 1), 1, False, 8, 4, 6, 10),
    ("t5", 128, 128, (5, 1, 1), 1, False, 6, 3, 8, 8),
    ("3x3x3", 64, 64, (3, 3, 3), 1, False, 6, 3, 10, 12),
    ("small_cin7", 7, 128, (1, 3, 3), 1, False, 4, 2, 20, 16),
    ("small_cin3_3x3x3", 3, 64, (3, 3, 3), 1, False, 3, 3, 12, 12),
    ("out4", 128, 4, (1, 3, 3), 1, False, 2, 2, 16, 16),
    ("out3", 128, 3, (1, 3, 3), 1, False, 2, 1, 16, 16),
    # (conv_out shapes: rows that are no multiple of 8 / 64, two images, every channel count of the UNet / VAE conv_out.  A vector-ALU kernel
    #  for these 4-channel convs was tried in round 6 — 1.08 against 1.15 ms at M = 1 638 400: both re-read the input once per tap from L2,
    #  which is what bounds them — and dropped)
    ("out4_c256_row_tails", 256, 4, (1, 3, 3), 1, False, 2, 1, 17, 19),
    ("out4_c64_small", 64, 4, (1, 3, 3), 1, False, 3, 3, 9, 11),
    ("out4_c128_wide", 128, 4, (1, 3, 3), 1, False, 2, 2, 40, 72),
    # cout % 256 == 0: also run by the 256x256-tile kernel when UAV_CONV_TILE=256 (or a large grid)
    ("big_3x3_c128_n256", 128, 256, (1, 3, 3), 1, False, 2, 2, 23, 17),
    ("big_t3_c256_n256", 256, 256, (3, 1, 1), 1, False, 6, 3, 9, 11),
    ("big_up_c64_n256", 64, 256, (1, 3, 3), 1, True, 2, 1, 9, 7),
    ("big_s2_c64_n512", 64, 512, (1, 3, 3), 2, False, 2, 2, 18, 14),
    ("big_1x1_c192_n512", 192, 512, (1, 1, 1), 1, False, 3, 1, 31, 9),
    # a frame is a whole number of 256-row tiles and the grid is large: the 256x256 kernel skips the temporal taps that
    # fall outside the 3-frame chunk for a whole tile (first / last frame) — decoder 3x3x3 conv and a (5,1,1) conv
    ("frames_3x3x3_c64_n256", 64, 256, (3, 3, 3), 1, False, 6, 3, 112, 96),
    ("frames_t5_c128_n256", 128, 256, (5, 1, 1), 1, False, 16, 8, 64, 64),
]


def ref_conv(x5, w, b, kt_khw, stride, upsample):
    """x5: (B,C,T,H,W) fp32; w: (O,I,kt,kh,kw)."""
    kt, kh, kw = kt_khw
    if upsample:
        x5 = F.interpolate(x5, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
    return F.conv3d(x5, w, b, stride=(1, stride, stride), padding=(kt // 2, kh // 2, kw // 2))


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_gemm(ops, dev, case):
    name, cin, cout, k3, stride, ups, n_img, t_len, h, w = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    bsz = n_img // t_len
    x5 = h16(bsz, cin, t_len, h, w, dev=dev, gen=g)
    fan = cin * k3[0] * k3[1] * k3[2]
    wt = h16(cout, cin, *k3, dev=dev, scale=fan ** -0.5, gen=g)
    bias = torch.randn(cout, generator=g).to(dev)
    ref = ref_conv(x5, wt, bias, k3, stride, ups)                      # (B,O,T,Ho,Wo)
    ho, wo = ref.shape[-2:]
    rows = x5.permute(0, 2, 3, 4, 1).reshape(-1, cin)
    if cin % 64:
        rows = F.pad(rows, (0, 8 - cin))
    rows = rows.contiguous().half()
    cw = ops.pack_conv(wt, bias, device=dev)
    out_f32 = cout < 8
    y = ops.conv_gemm(rows, cw, n_img=n_img, t_len=t_len, hi=h, wi=w, stride=stride, upsample=ups, out_f32=out_f32)
    y5 = y[:, :cout].float().reshape(bsz, t_len, ho, wo, cout).permute(0, 4, 1, 2, 3)
    err = rel_l2(y5, ref)
    assert err < (1e-4 if out_f32 else 2e-3), f"{name}: rel-L2 {err}"
    if cw.n > cout:                                                     # padded store column is exactly zero+0 bias
        assert y[:, cout:].abs().max().item() == 0.0


def test_conv_epilogue_fusions(ops, dev):
    """bias + per-batch row bias (temb) + residual + 1/output_scale_factor, two concatenated sources."""
    g = torch.Generator().manual_seed(7)
    bsz, t_len, h, w, c1, c2, cout = 2, 3, 10, 14, 128, 64, 256
    xa = h16(bsz, c1, t_len, h, w, dev=dev, gen=g); xb = h16(bsz, c2, t_len, h, w, dev=dev, gen=g)
    wt = h16(cout, c1 + c2, 1, 3, 3, dev=dev, scale=(9 * (c1 + c2)) ** -0.5, gen=g)
    bias = torch.randn(cout, generator=g).to(dev)
    temb = torch.randn(bsz, cout, generator=g).to(dev)
    res5 = h16(bsz, cout, t_len, h, w, dev=dev, gen=g)
    scale = 1.0 / 1.7
    ref = (ref_conv(torch.cat([xa, xb], 1), wt, bias, (1, 3, 3), 1, False) + temb[:, :, None, None, None] + res5) * scale
    ra = xa.permute(0, 2, 3, 4, 1).reshape(-1, c1).contiguous().half()
    rb = xb.permute(0, 2, 3, 4, 1).reshape(-1, c2).contiguous().half()
    rr = res5.permute(0, 2, 3, 4, 1).reshape(-1, cout).contiguous().half()
    cw = ops.pack_conv(wt, bias, device=dev)
    y = ops.conv_gemm(ra, cw, a2=rb, n_img=bsz * t_len, t_len=t_len, hi=h, wi=w, rowbias=temb.contiguous(),
                      rows_per_batch=t_len * h * w, residual=rr, out_scale=scale)
    y5 = y.float().reshape(bsz, t_len, h, w, cout).permute(0, 4, 1, 2, 3)
    assert rel_l2(y5, ref) < 2e-3


def test_linear_geglu_and_plain(ops, dev):
    g = torch.Generator().manual_seed(11)
    m, k, f = 1000, 128, 256
    x = h16(m, k, dev=dev, gen=g)
    w = h16(2 * f, k, dev=dev, scale=k ** -0.5, gen=g)
    b = torch.randn(2 * f, generator=g).to(dev)
    proj = x @ w.t() + b
    hid, gate = proj.chunk(2, dim=-1)
    ref = hid * F.gelu(gate)
    cw = ops.pack_conv(w, b, geglu=True, device=dev)
    y = ops.linear(x.half(), cw)
    assert y.shape == (m, f)
    assert rel_l2(y, ref) < 2e-3
    cw2 = ops.pack_conv(w, b, device=dev)
    res = h16(m, 2 * f, dev=dev, gen=g)
    y2 = ops.linear(x.half(), cw2, residual=res.half())
    assert rel_l2(y2, proj + res) < 2e-3


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c1,c2,groups,n_inst,rows,silu", [
    (256, 0, 32, 2, 3 * 10 * 12, True),       # 5-D flavour: instance = T*H*W rows
    (128, 64, 32, 2, 500, True),              # concat, groups of 6 channels straddle the seam
    (512, 0, 32, 6, 77, False),               # per-frame flavour, no activation
    (1024, 1024, 32, 1, 300, True),
    (64, 0, 32, 3, 1000, True),
])
def test_groupnorm(ops, dev, c1, c2, groups, n_inst, rows, silu):
    g = torch.Generator().manual_seed(c1 + c2)
    c = c1 + c2
    x = h16(n_inst * rows, c, dev=dev, gen=g) * 1.5 + 0.3
    x = x.half().float()
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev); beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    xr = x.reshape(n_inst, rows, c).permute(0, 2, 1)                    # (N, C, L)
    ref = F.group_norm(xr, groups, gamma, beta, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(-1, c)
    x1 = x[:, :c1].contiguous().half()
    x2 = x[:, c1:].contiguous().half() if c2 else None
    y = ops.groupnorm(x1, gamma, beta, n_inst=n_inst, rows_per_inst=rows, groups=groups, eps=1e-6, silu=silu, x2=x2)
    assert rel_l2(y, ref) < 1.5e-3


def test_groupnorm_padded_channels(ops, dev):
    """3 real channels padded to 8 (video-VAE condition branch: GroupNorm(3 groups, 3 channels))."""
    g = torch.Generator().manual_seed(3)
    rows = 640
    x = torch.zeros(rows, 8, device=dev)
    x[:, :3] = h16(rows, 3, dev=dev, gen=g)
    gamma = torch.tensor([1.1, 0.9, 1.3], device=dev); beta = torch.tensor([0.1, -0.2, 0.0], device=dev)
    ref = F.silu(F.group_norm(x[:, :3].t()[None], 3, gamma, beta, eps=1e-6))[0].t()
    y = ops.groupnorm(x.half(), gamma, beta, n_inst=1, rows_per_inst=rows, groups=3, eps=1e-6, silu=True, c_real=3)
    assert rel_l2(y[:, :3], ref) < 1.5e-3
    assert y[:, 3:].abs().max().item() == 0.0


@pytest.mark.parametrize("c", [128, 512, 1024])
def test_layernorm(ops, dev, c):
    g = torch.Generator().manual_seed(c)
    x = h16(777, c, dev=dev, gen=g) * 2 + 0.5
    x = x.half().float()
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev); beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    ref = F.layer_norm(x, (c,), gamma, beta, 1e-5)
    y = ops.layernorm(x.half(), gamma, beta, 1e-5)
    assert rel_l2(y, ref) < 1.5e-3


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,heads,bq,lq,lk,qpk", [
    (64, 8, 4, 300, 77, 2),        # text cross-attention, 2 frames per text
    (128, 8, 2, 200, 200, 1),      # spatial self-attention, ragged L
    (64, 2, 2, 130, 33, 1),
    (128, 8, 2, 200, 77, 1),       # text cross-attention at the 1024-channel level (short-key kernel, d = 128, ragged last block)
    (64, 8, 2, 1600, 77, 2),       # ... whole 128-query blocks
    (128, 2, 1, 1600, 1600, 1),
    (512, 1, 2, 160, 160, 1),      # VAE mid-block attention, single head
    (512, 1, 1, 100, 1000, 1),
    (512, 1, 2, 300, 2000, 1),     # lk >= 1024: the ring kernel on pre-packed K / V^T (csrc/attn512x.hip); ragged query block and key tile
    (512, 1, 3, 256, 4096, 1),
])
def test_attention(ops, dev, d, heads, bq, lq, lk, qpk):
    g = torch.Generator().manual_seed(d + lq)
    c = heads * d
    q = h16(bq, lq, c, dev=dev, gen=g)
    k = h16(bq // qpk, lk, c, dev=dev, gen=g)
    v = h16(bq // qpk, lk, c, dev=dev, gen=g)
    qh = q.reshape(bq, lq, heads, d).permute(0, 2, 1, 3)
    kh = k.repeat_interleave(qpk, 0).reshape(bq, lk, heads, d).permute(0, 2, 1, 3)
    vh = v.repeat_interleave(qpk, 0).reshape(bq, lk, heads, d).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1)
    ref = (p @ vh).permute(0, 2, 1, 3).reshape(bq * lq, c)
    out = ops.attention(q.half().reshape(-1, c), k.half().reshape(-1, c), v.half().reshape(-1, c), bq=bq, lq=lq, lk=lk,
                        heads=heads, head_dim=d, q_per_kv=qpk)
    assert rel_l2(out, ref) < 3e-3


def test_attention_fused_qkv_strides(ops, dev):
    """q/k/v as column slices of one fused [rows][3C] projection (row stride 3C)."""
    g = torch.Generator().manual_seed(5)
    heads, d, b, l = 2, 128, 2, 96
    c = heads * d
    qkv = h16(b * l, 3 * c, dev=dev, gen=g).half()
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    qh, kh, vh = [t.float().reshape(b, l, heads, d).permute(0, 2, 1, 3) for t in (q, k, v)]
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(b * l, c)
    out = ops.attention(q, k, v, bq=b, lq=l, lk=l, heads=heads, head_dim=d)
    assert rel_l2(out, ref) < 3e-3


def rope_tables(t_len, rot_dim, dev):
    freqs = 1.0 / (10000 ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    ang = torch.arange(t_len).float()[:, None] * freqs[None]
    return ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev), freqs


def ref_rope(x, freqs):
    """x: (..., T, d); rotate first 2*len(freqs) dims, interleaved pairs."""
    t = x.shape[-2]
    rd = freqs.numel() * 2
    ang = (torch.arange(t, device=x.device).float()[:, None] * freqs.to(x.device)[None]).repeat_interleave(2, -1)
    xr, xp = x[..., :rd], x[..., rd:]
    x1, x2 = xr[..., 0::2], xr[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(xr.shape)
    return torch.cat([xr * ang.cos() + rot * ang.sin(), xp], dim=-1)


@pytest.mark.parametrize("c,heads,t_len,hw,nb", [(512, 8, 8, 50, 2), (1024, 8, 5, 33, 1), (128, 2, 8, 40, 2), (256, 2, 3, 20, 1)])
def test_temporal_attention(ops, dev, c, heads, t_len, hw, nb):
    g = torch.Generator().manual_seed(c + t_len)
    d = c // heads
    qkv = h16(nb * t_len * hw, 3 * c, dev=dev, gen=g)
    bias = (0.5 * torch.randn(heads, t_len, t_len, generator=g)).to(dev).contiguous()
    cos, sin, freqs = rope_tables(t_len, 32, dev)
    scale = d ** -0.5
    x = qkv.reshape(nb, t_len, hw, 3, heads, d).permute(3, 0, 2, 4, 1, 5)       # (3, b, p, h, t, d)
    q, k, v = x[0] * scale, x[1], x[2]
    q = ref_rope(q, freqs); k = ref_rope(k, freqs)
    s = q @ k.transpose(-1, -2) + bias[None, None]
    s = s - s.amax(-1, keepdim=True)
    o = torch.softmax(s, -1) @ v                                                   # (b, p, h, t, d)
    ref = o.permute(0, 3, 1, 2, 4).reshape(nb * t_len * hw, c)
    out = ops.temporal_attention(qkv.half(), n_batch=nb, t_len=t_len, hw=hw, c=c, heads=heads, scale=scale,
                                 rope_cos=cos, rope_sin=sin, rot_dim=32, bias=bias)
    assert rel_l2(out, ref) < 2e-3


# ------------------------------------------------------------------------------------------------
def test_linear_small_and_timestep_embedding(ops, dev):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 256, generator=g).to(dev)
    w = h16(1024, 256, dev=dev, scale=1 / 16, gen=g); b = torch.randn(1024, generator=g).to(dev)
    y = ops.linear_small(x, w.half(), b, post_silu=True)
    assert rel_l2(y, F.silu(x @ w.t() + b)) < 1e-5
    y2 = ops.linear_small(y, h16(512, 1024, dev=dev, scale=1 / 32, gen=g).half(), None, pre_silu=True)
    assert y2.shape == (2, 512) and torch.isfinite(y2).all()
    t = torch.tensor([958.0, 1.0, 34.0], device=dev)
    emb = ops.timestep_embedding(t, 256, True, 0.0)
    half = 128
    expo = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / half
    arg = t[:, None] * torch.exp(expo)[None]
    ref = torch.cat([arg.cos(), arg.sin()], -1)          # flip_sin_to_cos=True -> [cos, sin]
    assert (emb - ref).abs().max().item() < 2e-4


def test_pack_unpack(ops, dev):
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 4, 3, 6, 10, generator=g).to(dev).half()
    b = torch.randn(2, 3, 3, 6, 10, generator=g).to(dev).half()
    rows = ops.pack_nhwc(a, b, c_pad=8)
    ref = torch.cat([a, b], 1).permute(0, 2, 3, 4, 1).reshape(-1, 7)
    assert torch.equal(rows[:, :7], ref) and rows[:, 7].abs().max().item() == 0
    back = ops.unpack_ncthw(rows, c=4, n_batch=2, t_len=3, h=6, w=10)
    assert torch.equal(back, a)
    rows32 = ops.pack_nhwc(a.float(), None, c_pad=8, scale=2.0)
    assert torch.equal(rows32[:, :4], (a.float() * 2).half().permute(0, 2, 3, 4, 1).reshape(-1, 4))
    cl = ops.unpack_ncthw(rows32.float().contiguous(), c=4, n_batch=2, t_len=3, h=6, w=10, out_dtype=torch.float32,
                          clamp=(-1.0, 1.0))
    assert cl.abs().max().item() <= 1.0


def test_cfg_ddim(ops, dev):
    g = torch.Generator().manual_seed(4)
    n = 4 * 8 * 33 * 17 + 3
    eu, ec, x = [torch.randn(n, generator=g).to(dev).half() for _ in range(3)]
    a, bcoef, gs = 0.83, -0.55, 6.0
    gd, x0 = ops.cfg_ddim_v0(eu, ec, x, guidance=gs, coef_sample=a, coef_eps=bcoef)
    gref = eu.float() + gs * (ec.float() - eu.float())
    assert rel_l2(gd, gref) < 1e-3
    assert rel_l2(x0, a * x.float() + bcoef * gd.float()) < 1e-3
    prev = ops.ddim_vt(x0, gd, x, coef_x0=0.9, coef_dir=0.4, eps_from_model=a, eps_from_sample=0.5)
    ref = 0.9 * x0.float() + 0.4 * (a * gd.float() + 0.5 * x.float())
    assert rel_l2(prev, ref) < 1e-3
    y = ops.axpby(eu, ec, 0.5, 0.5)
    assert rel_l2(y, 0.5 * eu.float() + 0.5 * ec.float()) < 1e-3


def ref_flow_warp(x, flow, mode):
    """flow_warp of the reference (propagation_module.py:104-135) in fp32."""
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h, device=x.device).float(), torch.arange(w, device=x.device).float(), indexing="ij")
    vx = gx + flow[:, 0]; vy = gy + flow[:, 1]
    grid = torch.stack((2.0 * vx / max(w - 1, 1) - 1.0, 2.0 * vy / max(h - 1, 1) - 1.0), dim=3)
    return F.grid_sample(x, grid, mode=mode, padding_mode="zeros", align_corners=True)


@pytest.mark.parametrize("nearest", [True, False])
def test_propagate_step_fp32_coords(ops, dev, nearest):
    g = torch.Generator().manual_seed(9)
    c, h, w = 4, 40, 56
    prev = torch.randn(1, c, h, w, generator=g).to(dev).half()
    cur = torch.randn(1, c, h, w, generator=g).to(dev).half()
    # flows with a sub-pixel offset of .3/.7 so nearest never sits on a .5 tie in fp32
    fp = (torch.randint(-3, 4, (1, 2, h, w), generator=g).float() + 0.3).to(dev).half()
    fc = (-fp.float() + 0.05 * torch.randn(1, 2, h, w, generator=g).to(dev)).half()
    fpf, fcf = fp.float(), fc.float()
    bw = ref_flow_warp(fcf, fpf, "bilinear")
    diff = fpf + bw
    lsq = lambda t: (t * t).sum(1, keepdim=True)
    valid = (lsq(diff) < 0.01 * (lsq(fpf) + lsq(bw)) + 0.5).float()
    warped = ref_flow_warp(prev.float(), fpf, "nearest" if nearest else "bilinear")
    fused = warped * 0.5 + cur.float() * 0.5
    ref = valid * fused + (1 - valid) * cur.float()
    out = torch.empty_like(cur[0])
    ops.propagate_step(prev[0], cur[0], fp[0], fc[0], out, c=c, h=h, w=w, feat_chan_stride=h * w, flow_chan_stride=h * w,
                       nearest=nearest, coord_f16=False, fuse_scale=0.5, alpha1=0.01, alpha2=0.5)
    bad = ((out.float() - ref[0]).abs() > 2e-2).float().mean().item()
    assert bad < 2e-3, f"{bad} of pixels differ"          # threshold-boundary pixels may flip the mask


# ------------------------------------------------------------------------------------------------
# fp32 residual stream of the VAE decoder (round 2): fp32 conv outputs / fp32 residual, GroupNorm on fp32 rows
@pytest.mark.parametrize("cout,h,w,tile", [(256, 40, 36, "128 or 256"), (128, 12, 10, "128, M tail -> generic path"),
                                           (512, 48, 48, "256x256 kernel")])
def test_conv_f32_stream_epilogue(ops, dev, cout, h, w, tile):
    """out_f32 with an fp32 residual: ((conv + bias) + residual) * scale, stored fp32.  Tolerance 1e-4 (fp32 output,
    fp16-representable operands, fp32 accumulation; only the summation order differs from the torch reference)."""
    g = torch.Generator().manual_seed(cout + h)
    bsz, t_len, cin = 2, 3, 128
    x5 = h16(bsz, cin, t_len, h, w, dev=dev, gen=g)
    wt = h16(cout, cin, 1, 3, 3, dev=dev, scale=(9 * cin) ** -0.5, gen=g)
    bias = torch.randn(cout, generator=g).to(dev)
    res5 = torch.randn(bsz, cout, t_len, h, w, generator=g).to(dev)                 # NOT fp16-representable on purpose
    scale = 1.0 / 1.3
    ref = (ref_conv(x5, wt, bias, (1, 3, 3), 1, False) + res5) * scale
    rows = x5.permute(0, 2, 3, 4, 1).reshape(-1, cin).contiguous().half()
    rr = res5.permute(0, 2, 3, 4, 1).reshape(-1, cout).contiguous()
    cw = ops.pack_conv(wt, bias, device=dev)
    y = ops.conv_gemm(rows, cw, n_img=bsz * t_len, t_len=t_len, hi=h, wi=w, residual=rr, out_scale=scale, out_f32=True)
    assert y.dtype == torch.float32
    y5 = y.reshape(bsz, t_len, h, w, cout).permute(0, 4, 1, 2, 3)
    assert rel_l2(y5, ref) < 1e-4
    y0 = ops.conv_gemm(rows, cw, n_img=bsz * t_len, t_len=t_len, hi=h, wi=w, out_f32=True)          # no residual
    assert rel_l2(y0.reshape(bsz, t_len, h, w, cout).permute(0, 4, 1, 2, 3), ref_conv(x5, wt, bias, (1, 3, 3), 1, False)) < 1e-4
    yh = ops.conv_gemm(rows, cw, n_img=bsz * t_len, t_len=t_len, hi=h, wi=w, residual=rr, out_scale=scale)   # fp32 res, fp16 out
    assert yh.dtype == torch.float16 and rel_l2(yh.float().reshape(bsz, t_len, h, w, cout).permute(0, 4, 1, 2, 3), ref) < 2e-3


def test_groupnorm_on_fp32_rows(ops, dev):
    g = torch.Generator().manual_seed(23)
    n_inst, rows_per, c, groups = 3, 700, 256, 32
    x = (torch.randn(n_inst * rows_per, c, generator=g) * 2.0 + 0.7).to(dev)       # fp32, not fp16-representable
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev); beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    y = ops.groupnorm(x, gamma, beta, n_inst=n_inst, rows_per_inst=rows_per, groups=groups, eps=1e-6, silu=True)
    xr = x.reshape(n_inst, rows_per, groups, c // groups).double()
    mu = xr.mean(dim=(1, 3), keepdim=True); var = xr.var(dim=(1, 3), unbiased=False, keepdim=True)
    ref = F.silu((((xr - mu) / (var + 1e-6).sqrt()).float().reshape(-1, c)) * gamma + beta)
    assert y.dtype == torch.float16 and rel_l2(y, ref) < 1e-3                      # one fp16 rounding of the output
    # two fp32 sources (channel concat)
    x2 = torch.randn(n_inst * rows_per, 64, generator=g).to(dev)
    g2 = torch.ones(c + 64, device=dev); b2 = torch.zeros(c + 64, device=dev)
    y2 = ops.groupnorm(x, g2, b2, n_inst=n_inst, rows_per_inst=rows_per, groups=32, eps=1e-6, silu=False, x2=x2)
    xc = torch.cat([x, x2], -1).reshape(n_inst, rows_per, 32, (c + 64) // 32).double()
    mu = xc.mean(dim=(1, 3), keepdim=True); var = xc.var(dim=(1, 3), unbiased=False, keepdim=True)
    assert rel_l2(y2, ((xc - mu) / (var + 1e-6).sqrt()).float().reshape(-1, c + 64)) < 1e-3


def test_cast_and_sft_fuse(ops, dev):
    g = torch.Generator().manual_seed(29)
    x = torch.randn(1003, 24, generator=g).to(dev)
    assert torch.equal(ops.cast_f16(x), x.half())
    assert ops.cast_f16(x.half()).dtype == torch.float16
    dec, sc, sh = [torch.randn(777, 64, generator=g).to(dev) for _ in range(3)]
    ref = dec + 0.6 * (dec * sc + sh)
    assert rel_l2(ops.sft_fuse(dec, sc, sh, 0.6, out_f32=True), ref) < 1e-6
    assert rel_l2(ops.sft_fuse(dec.half(), sc.half(), sh.half(), 0.6), dec.half().float() + 0.6 * (dec.half().float() * sc.half().float() + sh.half().float())) < 1e-3


def test_resize_area_and_propagation_flow_resize(ops, dev):
    """F.interpolate(mode='area') kernel (integer and fractional ratios) and Propagation.forward with flows at twice
    the latent resolution (reference propagation_module.py:206-209: area resize, flows scaled by w / w_f)."""
    g = torch.Generator().manual_seed(41)
    x = torch.randn(3, 2, 37, 50, generator=g)
    for (ho, wo) in ((37, 50), (18, 25), (12, 17), (74, 100)):
        ref = F.interpolate(x, (ho, wo), mode="area") * 0.5
        out = ops.resize_area_f32(x.to(dev), ho, wo, mul=0.5)
        assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() < 1e-6
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import golden_cases as GC
    from models_video.propagation_module import Propagation
    xl, ff, fb = GC.prop_inputs(5, 24, 32)
    up = lambda f: F.interpolate(f.reshape(1, 2 * 4, 24, 32), scale_factor=2, mode="nearest").reshape(1, 2, 4, 48, 64) * 2.0
    prop = Propagation(4, learnable=False); prop.coord_f16 = False
    a = prop(xl.half().to(dev), ff.to(dev), fb.to(dev), interpolation="nearest", mode="fuse", fuse_scale=0.5, alpha1=0.001, alpha2=0.05)
    b = prop(xl.half().to(dev), up(ff).to(dev), up(fb).to(dev), interpolation="nearest", mode="fuse", fuse_scale=0.5, alpha1=0.001, alpha2=0.05)
    assert ((a.float() - b.float()).abs() > 1e-2).float().mean().item() < 5e-3


# ------------------------------------------------------------------------------------------------
# GroupNorm statistics fused into the producing conv's epilogue (uav_conv_params.gn_partials)
@pytest.mark.parametrize("cout,f32,res,temb,k3", [
    (256, False, True, True, (1, 3, 3)),       # 8 channels per group: both half-waves of a quad pair
    (256, True, True, False, (1, 1, 1)),
    (512, False, True, True, (1, 1, 1)),       # 16
    (512, True, False, False, (1, 3, 3)),
    (1024, False, False, True, (3, 1, 1)),     # 32: one column tile per group
    (2048, False, True, False, (1, 1, 1)),     # 64: two column tiles per group
])
def test_conv_fused_groupnorm_statistics(ops, dev, cout, f32, res, temb, k3):
    """The partials a conv epilogue writes must (a) leave the conv output bit-identical, (b) equal the per-chunk sums of
    the stored values (tolerance 2e-3 relative to the chunk's L1 / L2 mass: they are taken before the fp16 rounding of
    the output), (c) give the same scale/shift tables as the stand-alone statistics pass to 1e-3, and (d) be picked up by
    ops.groupnorm through the tensor the conv returned."""
    g = torch.Generator().manual_seed(cout + 7 * k3[0])
    bsz, t_len, h, w, cin, groups = 2, 2, 64, 64, 64, 32         # enough 256x256 tiles (>= 224) for the big kernel at every width:
    if cout < 1024:
        h, w = (128, 128) if cout == 256 else (128, 64)           # M/256 x cout/256 = 256 tiles
    M = bsz * t_len * h * w
    x = (torch.randn(M, cin, generator=g)).half().to(dev)
    wt = h16(cout, cin, *k3, dev=dev, scale=(cin * k3[0] * k3[1] * k3[2]) ** -0.5, gen=g)
    bias = torch.randn(cout, generator=g).to(dev)
    cw = ops.pack_conv(wt, bias, device=dev)
    rr = None
    if res:
        rr = torch.randn(M, cout, generator=g).to(dev)
        rr = rr if f32 else rr.half()
    rb = torch.randn(bsz, cout, generator=g).to(dev).contiguous() if temb else None
    kw = dict(n_img=bsz * t_len, t_len=t_len, hi=h, wi=w, residual=rr, out_scale=1.0 / 1.3, out_f32=f32, rowbias=rb,
              rows_per_batch=t_len * h * w)
    y0 = ops.conv_gemm(x, cw, **kw)
    y = ops.conv_gemm(x, cw, gn_groups=groups, **kw)
    gn = getattr(y, "_uav_gn", None)
    assert gn is not None and gn.rows == 64 and gn.ws.shape == (2, groups, M // 64), "launch did not produce partials"
    assert torch.equal(y0, y)
    cpg = cout // groups
    yc = y.double().reshape(M // 64, 64, groups, cpg)
    s_ref = yc.sum(dim=(1, 3)).t(); q_ref = (yc * yc).sum(dim=(1, 3)).t()
    l1 = yc.abs().sum(dim=(1, 3)).t()
    es = ((gn.ws[0].double() - s_ref).abs() / (l1 + 1e-6)); eq = ((gn.ws[1].double() - q_ref).abs() / (q_ref + 1e-6))
    assert es.max().item() < 2e-3, (es.max().item(), (es > 2e-3).nonzero()[:8].tolist(), gn.ws[0][:2, :4].tolist(), s_ref[:2, :4].tolist())
    assert eq.max().item() < 2e-3, (eq.max().item(), (eq > 2e-3).nonzero()[:8].tolist())
    gamma = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev); beta = (0.1 * torch.randn(cout, generator=g)).to(dev)
    for n_inst, rows_per in ((bsz, t_len * h * w), (bsz * t_len, h * w)):        # 5-D and per-frame GroupNorm
        sc1, sh1 = ops.groupnorm_scale_shift(y, gamma, beta, n_inst=n_inst, rows_per_inst=rows_per, groups=groups, eps=1e-6)
        sc0, sh0 = ops.groupnorm_scale_shift(y0, gamma, beta, n_inst=n_inst, rows_per_inst=rows_per, groups=groups, eps=1e-6)
        assert rel_l2(sc1, sc0) < 1e-3 and rel_l2(sh1, sh0) < 1e-3
    # an in-place update of the tensor invalidates the partials (stand-alone pass runs instead: result follows the data)
    y.mul_(2.0)
    sc2, _ = ops.groupnorm_scale_shift(y, gamma, beta, n_inst=bsz, rows_per_inst=t_len * h * w, groups=groups, eps=1e-6)
    sc5, _ = ops.groupnorm_scale_shift(y0, gamma, beta, n_inst=bsz, rows_per_inst=t_len * h * w, groups=groups, eps=1e-6)
    assert rel_l2(sc2 * 2.0, sc5) < 1e-3


@pytest.mark.parametrize("c1,c2,f32,half_skip", [(512, 512, False, False), (256, 256, True, True), (1024, 1024, True, False),
                                                  (1024, 512, False, False)])
def test_groupnorm_two_source_statistics_from_producer_partials(ops, dev, c1, c2, f32, half_skip):
    """GroupNorm over a channel-concatenated input [x1 | x2] (up-block ResNets; the concat is never materialised): when the 32
    output groups do not straddle the seam, the scale/shift tables come from the partials the two PRODUCING convs wrote
    (uav_groupnorm_finalize_partials2: each output group = whole producer groups), incl. a skip tensor that exists once for
    both batch entries; 1024 + 512 (48-channel groups straddle) must fall back to the statistics pass."""
    g = torch.Generator().manual_seed(c1 + c2)
    bsz, t_len, h, w, cin, groups = 2, 2, 64, 64, 64, 32
    if c1 < 1024:
        h, w = (128, 128) if c1 == 256 else (128, 64)
    rows = bsz * t_len * h * w

    def produce(cout, nb):
        m = nb * t_len * h * w
        x = torch.randn(m, cin, generator=g).half().to(dev)
        cw = ops.pack_conv(h16(cout, cin, 1, 1, 1, dev=dev, scale=cin ** -0.5, gen=g), torch.randn(cout, generator=g), device=dev)
        return ops.conv_gemm(x, cw, n_img=nb * t_len, t_len=t_len, hi=h, wi=w, out_f32=f32, gn_groups=groups)
    x1 = produce(c1, bsz)
    x2 = produce(c2, 1 if half_skip else bsz)
    c = c1 + c2
    straddle = c1 % (c // groups) != 0
    if not straddle:
        if half_skip and getattr(x2, "_uav_gn", None) is None:
            pytest.skip("half-batch producer too small for the 256x256 kernel")
        assert getattr(x1, "_uav_gn", None) is not None and getattr(x2, "_uav_gn", None) is not None
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev); beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    kw = dict(n_inst=bsz, rows_per_inst=t_len * h * w, groups=groups, eps=1e-5)
    fused = ops._gn_two_source(x1, x2, groups, bsz, t_len * h * w)
    assert (fused is None) == straddle
    sc1, sh1 = ops.groupnorm_scale_shift(x1, gamma, beta, x2=x2, **kw)
    x1c, x2c = x1.clone(), x2.clone()                       # copies carry no partials: stand-alone statistics pass
    sc0, sh0 = ops.groupnorm_scale_shift(x1c, gamma, beta, x2=x2c, **kw)
    assert rel_l2(sc1, sc0) < 1e-3 and rel_l2(sh1, sh0) < 1e-3
    # and against exact statistics of the concatenated tensor
    x2f = torch.cat([x2, x2]) if half_skip else x2
    xc = torch.cat([x1.double(), x2f.double()], dim=-1).reshape(bsz, t_len * h * w, groups, c // groups)
    mu = xc.mean(dim=(1, 3)); var = xc.var(dim=(1, 3), unbiased=False)
    sc_ref = (gamma.double().reshape(1, groups, -1) / (var + 1e-5).sqrt()[:, :, None]).reshape(bsz, c)
    assert rel_l2(sc1, sc_ref.float()) < 1e-3


@pytest.mark.parametrize("c,cs,cout,h,w,hilo,f32", [(128, 192, 128, 24, 20, False, False), (256, 384, 256, 128, 128, True, True),
                                                     (512, 256, 512, 64, 64, True, False)])
def test_shortcut_conv_folded_into_conv2(ops, dev, c, cs, cout, h, w, hilo, f32):
    """uav_conv_params.a2_center_tap: `conv_shortcut(x) + conv2(h)` (resnet.py:286-292) as ONE implicit GEMM — 3x3 over the C
    channels of h plus the centre tap over the shortcut operand (optionally a [hi | lo] pair with repeated weights) — against
    the two separate convs, on the 128x128 kernel (small grid: multiplies the structural zeros) and on the 256x256 kernel
    (skips them), with the GroupNorm statistics riding along."""
    g = torch.Generator().manual_seed(c + cs)
    n_img, t_len = 4, 2
    m = n_img * h * w
    hh = torch.randn(m, c, generator=g).half().to(dev)
    x32 = (torch.randn(m, cs, generator=g) * 2.0).to(dev)
    w2 = h16(cout, c, 3, 3, dev=dev, scale=(9 * c) ** -0.5, gen=g); b2 = torch.randn(cout, generator=g).to(dev)
    ws = h16(cout, cs, 1, 1, dev=dev, scale=cs ** -0.5, gen=g); bs = torch.randn(cout, generator=g).to(dev)
    hi = x32.half()
    raw = torch.cat([hi, (x32 - hi.float()).half()], dim=1).contiguous() if hilo else hi
    cw = ops.pack_conv_with_shortcut(w2, b2, ws, bs, 2 if hilo else 1, device=dev)
    kw = dict(n_img=n_img, t_len=t_len, hi=h, wi=w, out_scale=1.0 / 1.2, out_f32=f32, rows_per_batch=t_len * h * w)
    y = ops.conv_gemm(hh, cw, a2=raw, a2_center=True, gn_groups=32, **kw)
    ref = (F.conv2d(hh.float().reshape(n_img, h, w, c).permute(0, 3, 1, 2), w2.float(), b2, padding=1)
           + F.conv2d((raw[:, :cs].float() + (raw[:, cs:].float() if hilo else 0)).reshape(n_img, h, w, cs).permute(0, 3, 1, 2), ws.float(), bs)) / 1.2
    ref = ref.permute(0, 2, 3, 1).reshape(m, cout)
    assert y.dtype == (torch.float32 if f32 else torch.float16)
    assert rel_l2(y, ref) < (2e-4 if f32 else 6e-4), rel_l2(y, ref)
    if hilo:                                    # the pair carries x to ~22 bits: closer to the fp32 operand than one fp16 rounding
        exact = (F.conv2d(hh.float().reshape(n_img, h, w, c).permute(0, 3, 1, 2), w2.float(), b2, padding=1)
                 + F.conv2d(x32.reshape(n_img, h, w, cs).permute(0, 3, 1, 2), ws.float(), bs)) / 1.2
        assert rel_l2(y, exact.permute(0, 2, 3, 1).reshape(m, cout)) < (2e-4 if f32 else 6e-4)
    # two separate launches (shortcut as its own conv, result as the residual) give the same numbers
    res = ops.conv_gemm(raw, ops.pack_conv(torch.cat([ws, ws], dim=1) if hilo else ws, bs, device=dev), n_img=n_img, t_len=t_len, hi=h, wi=w,
                        out_f32=f32, rows_per_batch=t_len * h * w)
    y2 = ops.conv_gemm(hh, ops.pack_conv(w2, b2, device=dev), residual=res, **kw)
    assert rel_l2(y, y2) < (1e-5 if f32 else 1.5e-3)
    if m * cout // (256 * 256) >= 224:          # big kernel: statistics partials present and consistent
        gn = getattr(y, "_uav_gn", None)
        assert gn is not None
        gamma = torch.ones(cout, device=dev); beta = torch.zeros(cout, device=dev)
        sc1, sh1 = ops.groupnorm_scale_shift(y, gamma, beta, n_inst=n_img // t_len, rows_per_inst=t_len * h * w, groups=32, eps=1e-6)
        sc0, sh0 = ops.groupnorm_scale_shift(y.clone(), gamma, beta, n_inst=n_img // t_len, rows_per_inst=t_len * h * w, groups=32, eps=1e-6)
        assert rel_l2(sc1, sc0) < 1e-3 and rel_l2(sh1, sh0) < 1e-3


def test_conv_fused_groupnorm_statistics_not_offered(ops, dev):
    """Launches that cannot produce partials (small grids -> 128x128 kernel, N tails, GEGLU, 48-channel groups) return a
    plain tensor, and GroupNorm falls back to its own statistics pass."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * 16 * 16, 64, generator=g).half().to(dev)
    cw = ops.pack_conv(h16(256, 64, 1, 3, 3, dev=dev, scale=0.05, gen=g), torch.zeros(256), device=dev)
    y = ops.conv_gemm(x, cw, n_img=2, t_len=1, hi=16, wi=16, gn_groups=32)
    assert getattr(y, "_uav_gn", None) is None
    x2 = torch.randn(4 * 128 * 64, 64, generator=g).half().to(dev)
    cw2 = ops.pack_conv(h16(1536, 64, 1, 1, 1, dev=dev, scale=0.1, gen=g), torch.zeros(1536), device=dev)
    y2 = ops.conv_gemm(x2, cw2, n_img=4, t_len=1, hi=128, wi=64, gn_groups=32)       # 48 channels per group
    assert getattr(y2, "_uav_gn", None) is None
    cw3 = ops.pack_conv(h16(128, 64, 1, 1, 1, dev=dev, scale=0.1, gen=g), torch.zeros(128), device=dev)
    y3 = ops.conv_gemm(x2, cw3, n_img=4, t_len=1, hi=128, wi=64, gn_groups=32)       # 128 outputs: n_pad = 128 -> 128x128 kernel
    assert getattr(y3, "_uav_gn", None) is None


# ------------------------------------------------------------------------------------------------
# "nearest 2x + 3x3 conv" as four 2x2 sub-pixel phase convs with strided output rows (uav_conv_params.out_map_*)
@pytest.mark.parametrize("cin,cout,n_img,t_len,h,w,f32", [(64, 128, 2, 1, 9, 11, False), (128, 256, 4, 2, 48, 40, False),
                                                           (256, 256, 6, 3, 64, 64, True)])
def test_upsample_as_subpixel_phase_convs(ops, dev, cin, cout, n_img, t_len, h, w, f32):
    """Same result as the fused-gather upsampling conv (rel-L2 < 1e-3: only the fp16 rounding of the summed taps differs)
    and as the torch reference F.interpolate(nearest) + conv2d (< 2e-3 fp16 / 1e-3 fp32 rows out); small, tail and
    256x256-tile shapes."""
    g = torch.Generator().manual_seed(cin + h)
    x4 = h16(n_img, cin, h, w, dev=dev, gen=g)
    wt = h16(cout, cin, 3, 3, dev=dev, scale=(9 * cin) ** -0.5, gen=g)
    bias = torch.randn(cout, generator=g).to(dev)
    ref = F.conv2d(F.interpolate(x4, scale_factor=2.0, mode="nearest"), wt, bias, padding=1)
    rows = to_rows(x4)
    fused = ops.conv_gemm(rows, ops.pack_conv(wt[:, :, None], bias, device=dev), n_img=n_img, t_len=t_len, hi=h, wi=w,
                          upsample=True, out_f32=f32)
    ph = ops.upsample_phase_weights(wt)
    out = torch.full((n_img * 4 * h * w, cout), float("nan"), dtype=torch.float32 if f32 else torch.float16, device=dev)
    for py in range(2):
        for px in range(2):
            ops.conv_gemm(rows, ops.pack_conv(ph[py][px], bias, device=dev), n_img=n_img, t_len=t_len, hi=h, wi=w,
                          pad=(0, 1 - py, 1 - px), out_hw=(h, w), out_f32=f32, out=out, out_map=(w, 4 * w, 2, py * 2 * w + px))
    assert bool(torch.isfinite(out).all()), "a phase left output rows unwritten"
    y = from_rows(out, n_img, 2 * h, 2 * w)
    assert rel_l2(y, ref) < (1e-3 if f32 else 2e-3)
    assert rel_l2(out, fused) < 1e-3


@pytest.mark.parametrize("f32,per_frame", [(True, False), (True, True), (False, False)])
def test_upsampler_phase_convs_share_one_statistics_workspace(ops, dev, f32, per_frame):
    """Round 4: the four sub-pixel phase launches of `Upsample3D` write their GroupNorm partials into ONE workspace
    (uav_conv_params.gn_chunk_*: every output frame owns a contiguous run of chunks), so the GroupNorm that reads the up-sampled
    tensor needs no statistics pass: scale / shift from the shared partials == those of the stand-alone pass over the same
    tensor (fp64 finalize on both sides; 5-D instance and per-frame instance), and the partials ride on the returned tensor."""
    from uav import engine as E
    from models_video.resnet import Upsample3D
    g = torch.Generator().manual_seed(77)
    cout = 512
    up = Upsample3D(256, use_conv=True, out_channels=cout).to(dev)
    with torch.no_grad():
        up.conv.weight.copy_(h16(cout, 256, 3, 3, dev=dev, scale=(9 * 256) ** -0.5, gen=g))
        up.conv.bias.copy_(torch.randn(cout, generator=g).to(dev))
    geom = E.Geom(2, 4, 64, 64)                          # 8 frames of 64x64 -> 128x128: 128 x 2 tiles per phase -> the 256x256 kernel
    x = torch.randn(geom.rows, 256, generator=g).to(dev)
    x = x if f32 else x.half()
    saved = E.SAMPLER_HILO
    E.SAMPLER_HILO = False
    try:
        y, g2 = up.run(x, geom)
    finally:
        E.SAMPLER_HILO = saved
    gn = getattr(y, "_uav_gn", None)
    assert gn is not None and gn.filled == 4 and gn.ws.shape == (2, 32, g2.rows // 64), "the phase launches did not write the shared partials"
    gamma = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev); beta = (0.1 * torch.randn(cout, generator=g)).to(dev)
    n_inst, rpi = (g2.n_img, g2.hw) if per_frame else (g2.b, g2.rows_per_batch)
    kw = dict(n_inst=n_inst, rows_per_inst=rpi, groups=32, eps=1e-6)
    sc_f, sh_f = ops.groupnorm_scale_shift(y, gamma, beta, **kw)                  # from the shared partials
    y2 = y.clone()                                                                # a plain tensor: stand-alone statistics pass
    assert getattr(y2, "_uav_gn", None) is None
    sc_s, sh_s = ops.groupnorm_scale_shift(y2, gamma, beta, **kw)
    assert rel_l2(sc_f, sc_s) < 1e-6 and (sh_f - sh_s).abs().max().item() < 1e-5 * (1 + sh_s.abs().max().item())
    # and against torch on the NCHW view
    y4 = from_rows(y.float(), g2.n_img, g2.h, g2.w)                               # (B*T, C, H, W)
    y4 = y4 if per_frame else y4.reshape(g2.b, g2.t, cout, g2.h, g2.w).permute(0, 2, 1, 3, 4)
    ref = F.group_norm(y4.float(), 32, gamma, beta, eps=1e-6)
    rows = ops.groupnorm(y, gamma, beta, silu=False, **kw).float()
    out4 = from_rows(rows, g2.n_img, g2.h, g2.w)
    ref4 = ref if per_frame else ref.permute(0, 2, 1, 3, 4).reshape(g2.n_img, cout, g2.h, g2.w)
    assert rel_l2(out4, ref4) < 2e-3


# ------------------------------------------------------------------------------------------------
# second source read batch-broadcast (uav_conv_params.a2_images, x2_rows of the GroupNorm entry points)
@pytest.mark.parametrize("h,w,k3", [(48, 40, (1, 1, 1)), (24, 20, (1, 3, 3)), (160, 160, (1, 1, 1)), (128, 128, (1, 3, 3))])
def test_second_source_batch_broadcast(ops, dev, h, w, k3):
    """A skip tensor that exists once for both batch entries gives bit-identical results to its duplicated copy:
    GroupNorm statistics + apply over (x | skip) and a conv over the two sources (128x128 and 256x256 tile kernels)."""
    g = torch.Generator().manual_seed(h + k3[2])
    bsz, t_len, c1, c2, cout = 2, 2, 128, 64, 256
    rows = bsz * t_len * h * w
    x = torch.randn(rows, c1, generator=g).half().to(dev)
    skip1 = torch.randn(rows // 2, c2, generator=g).half().to(dev)
    skip2 = torch.cat([skip1, skip1])
    gamma = (1 + 0.1 * torch.randn(c1 + c2, generator=g)).to(dev); beta = (0.1 * torch.randn(c1 + c2, generator=g)).to(dev)
    kw = dict(n_inst=bsz, rows_per_inst=t_len * h * w, groups=32, eps=1e-6, silu=True)
    assert torch.equal(ops.groupnorm(x, gamma, beta, x2=skip1, **kw), ops.groupnorm(x, gamma, beta, x2=skip2, **kw))
    wt = h16(cout, c1 + c2, *k3, dev=dev, scale=((c1 + c2) * k3[1] * k3[2]) ** -0.5, gen=g)
    cw = ops.pack_conv(wt, torch.randn(cout, generator=g), device=dev)
    ckw = dict(n_img=bsz * t_len, t_len=t_len, hi=h, wi=w)
    assert torch.equal(ops.conv_gemm(x, cw, a2=skip1, **ckw), ops.conv_gemm(x, cw, a2=skip2, **ckw))


# ------------------------------------------------------------------------------------------------
# LayerNorm folded into the consuming projection (uav_conv_params.ln_*)
@pytest.mark.parametrize("k,n,geglu,res", [(512, 512, False, True), (512, 1536, False, False), (512, 4096, True, True), (1024, 1024, False, True)])
def test_layernorm_folded_into_projection(ops, dev, k, n, geglu, res):
    """A linear with an fp32 result (ln_produce) writes the fp16 operand copy + per-row statistics; the consuming projection
    (ln_consume, optionally GEGLU) applies rstd * (x16 . (W o gamma)^T - mu * colsum) + (W.beta + b) in its epilogue.  Against
    fp32 LayerNorm -> Linear (-> GEGLU) on the same rows, and against the unfused engine path (layernorm kernel + linear)."""
    import torch.nn as nn
    from uav import engine as E, _lib
    if not _lib.load().uav_has_dev_kernels():
        pytest.skip("the LayerNorm-fold conv instances are development kernels (measured slower and outside the parity bar, DESIGN section 6): "
                    "tools/ab/build_dev.sh all + UAV_HIP_LIB=tools/ab/libuav_hip_dev.so")
    g = torch.Generator().manual_seed(k + n)
    m = 128 * 512                                   # 256 m-tiles x >= 2 n-tiles: the 256x256 kernel
    x0 = torch.randn(m, k, generator=g).half().to(dev)
    wp = ops.pack_conv(h16(k, k, dev=dev, scale=k ** -0.5, gen=g), torch.randn(k, generator=g) * 0.1, device=dev)
    rr = (torch.randn(m, k, generator=g) * 1.5 + 0.3).to(dev) if res else None
    x = ops.linear(x0, wp, residual=rr, out_f32=True, ln_produce=True)          # the stream rows (fp32) + their LnOperand
    op = ops.ln_operand_of(x)
    assert op is not None and op.raw.shape == (m, k) and op.stat.shape == (k // 128, m, 2)
    assert torch.equal(op.raw, x.half())
    xc = x.double().reshape(m, k // 128, 128)
    assert rel_l2(op.stat[:, :, 0].t(), xc.sum(-1).float()) < 1e-5 and rel_l2(op.stat[:, :, 1].t(), (xc * xc).sum(-1).float()) < 1e-5
    assert torch.equal(x, ops.linear(x0, wp, residual=rr, out_f32=True))        # the fp32 rows themselves are unchanged

    class M(E.EngineModule):
        pass
    mod = M()
    ln = nn.LayerNorm(k).to(dev)
    lin = nn.Linear(k, n).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(k, generator=g)); ln.bias.copy_(0.1 * torch.randn(k, generator=g))
        lin.weight.copy_(h16(n, k, dev=dev, scale=k ** -0.5, gen=g).float()); lin.bias.copy_(0.1 * torch.randn(n, generator=g))
    old = E.LN_FOLD
    E.LN_FOLD = True                                # off by default (parity margin over 30 steps): the feature stays tested
    try:
        y = E.ln_linear(mod, "t", ln, x, [lin], geglu=geglu)
    finally:
        E.LN_FOLD = old
    with torch.no_grad():
        ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (k,), ln.weight, ln.bias, ln.eps), lin.weight, lin.bias)
        if geglu:
            a, b = ref.chunk(2, dim=-1)
            ref = a * torch.nn.functional.gelu(b)
    assert y.dtype == torch.float16 and y.shape == ref.shape
    e_fold = rel_l2(y, ref)
    E.LN_FOLD = False
    try:
        y0 = E.ln_linear(mod, "t", ln, x, [lin], geglu=geglu)
    finally:
        E.LN_FOLD = old
    e_plain = rel_l2(y0, ref)
    assert not torch.equal(y, y0)                   # the two paths really are different kernels
    assert e_fold < 1.5e-3 and e_fold < 2.0 * e_plain + 1e-4, (e_fold, e_plain)


# ------------------------------------------------------------------------------------------------
# Short-K kernel (round 5, conv_gemm_sk_kernel: 128 x 256 tile, two workgroups per CU) against the general 256x256 kernel
# (UAV_CONV_NO_SHORTK) and an fp32 reference — the linears / 1x1 convs of attention.py:523-564, resnet.py:286-292
SHORTK_CASES = [
    # name, c1, c2, cout, M, epilogue dict
    ("to_out_512_f32res", 512, 0, 512, 51200, dict(res="f32", out_f32=True)),
    ("proj_in_512_f32", 512, 0, 512, 28672 + 64 * 5, dict(out_f32=True, gn=32)),
    ("to_q_512_f16", 512, 0, 512, 30000, dict()),                                     # M tail: 30000 % 128 = 48
    ("short_256_res16_gn", 256, 0, 256, 57600, dict(res="f16", gn=32)),
    ("short_cat_768_256", 512, 256, 256, 57600 + 37, dict(res="f32", out_f32=True)),  # two sources, ragged M
    ("mid_1024_gn", 1024, 0, 1024, 25600, dict(res="f16", gn=32, scale=1.0 / 1.4)),
    ("qkv_512_1536", 512, 0, 1536, 30000, dict(bias=False)),
    ("geglu_512_4096", 512, 0, 4096, 25600, dict(geglu=True)),
    ("rowbias_f32_gn", 256, 0, 512, 2 * 16384, dict(rowbias=True, out_f32=True, gn=32)),
    ("rowbias_f16", 128, 0, 512, 2 * 16384, dict(rowbias=True, res="f16")),
    ("k64_two_stages", 64, 0, 256, 57600, dict(res="f16")),
    ("k128_res32_f16", 128, 0, 512, 28800, dict(res="f32")),
    ("bcast_a2", 256, 256, 512, 2 * 16000, dict(a2_half=True, res="f16")),
]


@pytest.mark.parametrize("case", SHORTK_CASES, ids=[c[0] for c in SHORTK_CASES])
def test_shortk_kernel_bit_identical_to_general_kernel(ops, dev, case):
    """Same tile-independent per-accumulator K order -> the short-K kernel must reproduce the 256x256 kernel BIT FOR BIT
    (outputs and fused GroupNorm partials), and both sit inside the conv tolerance against an fp32 reference."""
    name, c1, c2, cout, M, e = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    cin = c1 + c2
    geglu = e.get("geglu", False)
    n_out = cout // 2 if geglu else cout
    x1 = torch.randn(M, c1, generator=g).half().to(dev)
    x2 = None
    if c2:
        rows2 = M // 2 if e.get("a2_half") else M
        x2 = torch.randn(rows2, c2, generator=g).half().to(dev)
    wt = h16(cout, cin, dev=dev, scale=cin ** -0.5, gen=g)
    bias = torch.randn(cout, generator=g).to(dev) if e.get("bias", True) else None
    cw = ops.pack_conv(wt.reshape(cout, cin, 1, 1, 1), bias, geglu=geglu, device=dev)
    assert cw.k_pad == cin and cw.n_pad % 256 == 0
    res = None
    if e.get("res"):
        res = torch.randn(M, n_out, generator=g).to(dev)
        res = res if e["res"] == "f32" else res.half()
    nb = 2
    rb = torch.randn(nb, cout, generator=g).to(dev).contiguous() if e.get("rowbias") else None
    n_img, hi = ops._factor_rows(M) if not (rb is not None or e.get("a2_half")) else (nb, M // nb)
    kw = dict(a2=x2, n_img=n_img, t_len=1, hi=hi, wi=1, residual=res, out_scale=e.get("scale", 1.0), out_f32=e.get("out_f32", False),
              rowbias=rb, rows_per_batch=M // nb if rb is not None else 0, gn_groups=e.get("gn"))
    y_gen = ops.conv_gemm(x1, cw, no_shortk=True, no_w4=True, **kw)      # the 8-wave 256x256 kernel
    y_sk = ops.conv_gemm(x1, cw, **kw)                                   # default dispatch: four-wave kernel (UAV_CONV_SK=1: short-K kernel)
    assert torch.equal(y_gen, y_sk), (name, (y_gen.float() - y_sk.float()).abs().max().item())
    if e.get("gn"):
        g0, g1 = getattr(y_gen, "_uav_gn", None), getattr(y_sk, "_uav_gn", None)
        assert g0 is not None and g1 is not None and g0.ws.shape == g1.ws.shape
        # default dispatch = the four-wave kernel: its LDS epilogues sum the same fp32 values in another fixed order (see
        # test_w4_kernel_bit_identical_to_8wave_kernel); the short-K kernel (UAV_CONV_SK=1) shares the 8-wave epilogues: equal bits
        if os.environ.get("UAV_CONV_SK", "0") not in ("", "0"):
            assert torch.equal(g0.ws, g1.ws)
        else:
            assert torch.allclose(g0.ws, g1.ws, rtol=2e-5, atol=2e-4), (name, (g0.ws - g1.ws).abs().max().item())
    # fp32 reference on a row sample (incl. the last rows)
    idx = torch.cat([torch.arange(0, M, 97, device=dev), torch.arange(max(0, M - 300), M, device=dev)])
    xs = x1[idx].float()
    if c2:
        i2 = idx % x2.shape[0] if e.get("a2_half") else idx
        xs = torch.cat([xs, x2[i2].float()], dim=1)
    ref = xs @ wt.t()
    if bias is not None:
        ref = ref + bias
    if geglu:
        hid, gate = ref.chunk(2, dim=-1)
        ref = hid * F.gelu(gate)
    if rb is not None:
        ref = ref + rb[(idx // (M // nb))]
    if res is not None:
        ref = ref + res[idx].float()
    ref = ref * e.get("scale", 1.0)
    err = rel_l2(y_sk[idx], ref)
    assert err < (1e-4 if e.get("out_f32") else 2e-3), (name, err)


# ------------------------------------------------------------------------------------------------
# Four-wave kernel (round 5, conv_gemm256w_kernel: one wave per SIMD, 128 x 128 wave tiles, buffer-load gather with the
# hardware's out-of-range zero fill for padding) against the 8-wave kernel (UAV_CONV_NO_W4) — resnet.py:200-294, 94-196
W4_CASES = [
    # name, c1, c2, cout, (kt,kh,kw), stride, n_img, t_len, h, w, epilogue
    ("3x3_256_res32_gn", 256, 0, 256, (1, 3, 3), 1, 4, 2, 120, 128, dict(res="f32", out_f32=True, gn=32)),
    ("3x3_128_512_rowbias_gn", 128, 0, 512, (1, 3, 3), 1, 4, 2, 96, 80, dict(rowbias=True, gn=32)),
    ("3x3_cat_ragged", 128, 64, 256, (1, 3, 3), 1, 6, 3, 97, 101, dict(res="f16")),
    ("3x3_stride2", 128, 0, 512, (1, 3, 3), 2, 4, 2, 179, 181, dict(out_f32=True)),
    ("t3_256", 256, 0, 256, (3, 1, 1), 1, 16, 8, 64, 64, dict(res="f16", gn=32)),
    ("t5_128_ragged", 128, 0, 256, (5, 1, 1), 1, 12, 4, 70, 71, dict()),
    ("3x3x3_64_chunk3", 64, 0, 256, (3, 3, 3), 1, 6, 3, 112, 96, dict(res="f32", out_f32=True)),
    ("2x2_phase_like", 128, 0, 256, (1, 2, 2), 1, 4, 2, 128, 120, dict(pad=(0, 1, 1), out_hw=(128, 120))),
    ("1x1_k4608", 4608, 0, 512, (1, 1, 1), 1, 1, 1, 30000, 1, dict()),
    ("1x1_512_res32", 512, 0, 512, (1, 1, 1), 1, 1, 1, 51200, 1, dict(res="f32", out_f32=True, gn=32)),
    ("1x1_bcast_a2", 256, 256, 512, (1, 1, 1), 1, 2, 1, 16000, 1, dict(a2_half=True, res="f16")),
    ("3x3_bcast_a2", 128, 128, 256, (1, 3, 3), 1, 4, 2, 128, 128, dict(a2_half=True)),
    ("geglu_512_4096", 512, 0, 4096, (1, 1, 1), 1, 1, 1, 25600, 1, dict(geglu=True)),
    # row-coalesced fp32 epilogue (conv_epilogue_f32_lds): every statistics granularity (4 / 8 / 16 / 32 / 64 / 128 channels per group),
    # time-embedding row, plain, strided output rows
    ("3x3_256_rowbias_f32_gn8", 256, 0, 256, (1, 3, 3), 1, 4, 2, 128, 128, dict(rowbias=True, out_f32=True, gn=32)),
    ("3x3_256_f32_gn4", 256, 0, 256, (1, 3, 3), 1, 4, 2, 128, 128, dict(out_f32=True, gn=64)),
    ("1x1_1024_1024_res32_gn32", 1024, 0, 1024, (1, 1, 1), 1, 1, 1, 25600, 1, dict(res="f32", out_f32=True, gn=32)),
    ("1x1_2048_512_res32_gn64", 2048, 0, 512, (1, 1, 1), 1, 1, 1, 32768, 1, dict(res="f32", out_f32=True, gn=8)),
    ("1x1_2048_512_f32_gn128", 2048, 0, 512, (1, 1, 1), 1, 1, 1, 32768, 1, dict(out_f32=True, gn=4)),
    ("2x2_phase_like_f32", 256, 0, 256, (1, 2, 2), 1, 4, 2, 128, 120, dict(pad=(0, 1, 1), out_hw=(128, 120), out_f32=True)),
    # fp16 results through LDS (conv_epilogue_f16_lds / conv_epilogue_geglu_lds): fp32 residual -> fp16 operand (block tails), every
    # statistics granularity, no bias, K = 512 linears (the four-wave kernel takes every big-tile launch since run 25)
    ("1x1_512_res32_f16_gn16", 512, 0, 512, (1, 1, 1), 1, 1, 1, 51200, 1, dict(res="f32", gn=32)),
    ("3x3_256_f16_gn4", 256, 0, 256, (1, 3, 3), 1, 4, 2, 128, 128, dict(gn=64)),
    ("1x1_1024_1024_res16_gn32", 1024, 0, 1024, (1, 1, 1), 1, 1, 1, 25600, 1, dict(res="f16", gn=32)),
    ("1x1_2048_512_f16_gn128", 2048, 0, 512, (1, 1, 1), 1, 1, 1, 32768, 1, dict(gn=4)),
    ("1x1_512_512_nobias_res16", 512, 0, 512, (1, 1, 1), 1, 1, 1, 40000, 1, dict(res="f16", nobias=True)),
    ("1x1_512_1536_qkv", 512, 0, 1536, (1, 1, 1), 1, 1, 1, 30000, 1, dict()),
    ("geglu_512_4096_nobias", 512, 0, 4096, (1, 1, 1), 1, 1, 1, 12800, 1, dict(geglu=True, nobias=True)),
    ("1x1_512_rowbias_res16", 512, 0, 512, (1, 1, 1), 1, 4, 2, 128, 128, dict(rowbias=True, res="f16")),
]


@pytest.mark.parametrize("case", W4_CASES, ids=[c[0] for c in W4_CASES])
def test_w4_kernel_bit_identical_to_8wave_kernel(ops, dev, case):
    """Same LDS image, MFMA order and epilogues -> the four-wave kernel must reproduce the 8-wave kernel BIT FOR BIT (outputs
    and fused GroupNorm partials) on every tap set, stride, source split and tile tail — AND it is held directly to an fp32
    F.conv3d / GEMM reference of the same op (round 6), like the 8-wave kernel is by the tests above."""
    name, c1, c2, cout, k3, stride, n_img, t_len, h, w, e = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    cin = c1 + c2
    geglu = e.get("geglu", False)
    n_out = cout // 2 if geglu else cout
    rows = n_img * h * w
    x1 = torch.randn(rows, c1, generator=g).half().to(dev)
    x2 = None
    if c2:
        x2 = torch.randn(rows // 2 if e.get("a2_half") else rows, c2, generator=g).half().to(dev)
    wt = h16(cout, cin, *k3, dev=dev, scale=(cin * k3[0] * k3[1] * k3[2]) ** -0.5, gen=g)
    bias = None if e.get("nobias") else torch.randn(cout, generator=g).to(dev)
    cw = ops.pack_conv(wt, bias, geglu=geglu, device=dev)
    pad = e.get("pad", (k3[0] // 2, k3[1] // 2, k3[2] // 2))
    if e.get("out_hw"):
        ho, wo = e["out_hw"]
    else:
        ho = (h + 2 * pad[1] - k3[1]) // stride + 1
        wo = (w + 2 * pad[2] - k3[2]) // stride + 1
    M = n_img * ho * wo
    res = None
    if e.get("res"):
        res = torch.randn(M, n_out, generator=g).to(dev)
        res = res if e["res"] == "f32" else res.half()
    nb = n_img // t_len
    rb = torch.randn(nb, cout, generator=g).to(dev).contiguous() if e.get("rowbias") else None
    kw = dict(a2=x2, n_img=n_img, t_len=t_len, hi=h, wi=w, stride=stride, pad=pad, out_hw=e.get("out_hw"), residual=res,
              out_f32=e.get("out_f32", False), rowbias=rb, rows_per_batch=M // nb if rb is not None else 0, gn_groups=e.get("gn"),
              out_scale=1.0 / 1.3)
    y8 = ops.conv_gemm(x1, cw, no_w4=True, no_shortk=True, **kw)
    y4 = ops.conv_gemm(x1, cw, **kw)
    assert y8.shape == (M, n_out)
    diff = (y8.float() - y4.float()).abs()
    assert torch.equal(y8, y4), (name, diff.max().item(), (diff > 0).float().mean().item(), (diff > 0).nonzero()[:6].tolist())
    if e.get("gn"):
        g0, g1 = getattr(y8, "_uav_gn", None), getattr(y4, "_uav_gn", None)
        assert g0 is not None and g1 is not None
        # the four-wave kernel's results leave through its row-coalesced LDS epilogues (conv_epilogue_f32_lds / _f16_lds): the stored
        # VALUES are bit-identical (asserted above), the GroupNorm partials are sums of the same fp32 values in another, fixed order
        # (rows first, then the quads / octets of a group) — equal up to fp32 summation order: 64 x cpg terms per partial
        assert g0.ws.shape == g1.ws.shape
        assert torch.allclose(g0.ws, g1.ws, rtol=2e-5, atol=2e-4), (name, (g0.ws - g1.ws).abs().max().item())
        y4b = ops.conv_gemm(x1, cw, **kw)                                        # ... and deterministic: the same bits on a second launch
        assert torch.equal(getattr(y4b, "_uav_gn").ws, g1.ws)
    # ... and the four-wave kernel DIRECTLY against an fp32 reference of the same op (the kernel that runs 76 % of
    # the GPU time is not only held to the 8-wave kernel): F.conv3d / GEMM in fp32 on the same fp16-representable inputs, every output
    # element; < 2e-3 for fp16 results (2^-11 storage rounding), < 1e-4 for fp32 results
    xs = x1.float()
    if c2:
        xs = torch.cat([xs, (x2.repeat(2, 1) if e.get("a2_half") else x2).float()], dim=1)
    if k3 == (1, 1, 1) and stride == 1:
        ref = xs @ wt.reshape(cout, cin).t()
    else:
        x5 = xs.reshape(nb, t_len, h, w, cin).permute(0, 4, 1, 2, 3)
        tf32 = torch.backends.cudnn.allow_tf32
        torch.backends.cudnn.allow_tf32 = False
        try:
            ref5 = F.conv3d(x5, wt, None, stride=(1, stride, stride), padding=pad)
        finally:
            torch.backends.cudnn.allow_tf32 = tf32
        ref = ref5[..., :ho, :wo].permute(0, 2, 3, 4, 1).reshape(M, cout)
    if bias is not None:
        ref = ref + bias
    if rb is not None:
        ref = ref + rb[torch.arange(M, device=dev) // (M // nb)]
    if geglu:
        hid, gate = ref.chunk(2, dim=-1)
        ref = hid * F.gelu(gate)
    if res is not None:
        ref = ref + res.float()
    ref = ref * (1.0 / 1.3)
    err = rel_l2(y4[:, :n_out], ref)
    assert err < (1e-4 if e.get("out_f32") else 2e-3), (name, err)


# ------------------------------------------------------------------------------------------------
# Fused text cross-attention sub-layer (round 6, csrc/xattn_fused.hip): LayerNorm -> to_q -> 77-key softmax -> to_out -> + residual
# in one launch, against (a) the four-launch chain it replaces (same roundings: LayerNorm output, Q, P, O in fp16 -> equal up to fp32
# summation order and the fp16 roundings that order flips) and (b) an fp32 reference — reference attention.py:523-564 (attn1 with
# only_cross_attention / attn2), CrossAttention.forward :177-238
XATTN_CASES = [
    # name, kv batches, rows per kv batch, text keys, row mean / spread of x
    ("b2_1152_lk77", 2, 1152, 77, 0.3, 1.5),
    ("b1_128_lk77_one_tile", 1, 128, 77, 0.0, 1.0),
    ("b3_384_lk20_one_key_tile", 3, 384, 20, -2.0, 0.7),
    ("b2_25600_lk96_many_tiles", 2, 25600, 96, 5.0, 2.0),
]


@pytest.mark.parametrize("case", XATTN_CASES, ids=[c[0] for c in XATTN_CASES])
def test_fused_cross_attention_sublayer(ops, dev, case):
    name, nb, rpk, lk, mu, sd = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    C, H, D = 512, 8, 64
    M = nb * rpk
    x = (torch.randn(M, C, generator=g) * sd + mu + torch.randn(M, 1, generator=g)).to(dev)          # rows with their own offsets
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    wq = h16(C, C, dev=dev, scale=C ** -0.5, gen=g); wo = h16(C, C, dev=dev, scale=C ** -0.5, gen=g)
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    kv = (torch.randn(nb * lk, 2 * C, generator=g) * 1.5).half().to(dev)
    k, v = kv[:, :C], kv[:, C:]
    scale = D ** -0.5
    # (a) the chain
    n = ops.layernorm(x, gamma, beta, 1e-5)
    q = ops.linear(n, ops.pack_conv(wq, None, device=dev))
    o = ops.attention(q, k, v, bq=nb, lq=rpk, lk=lk, heads=H, head_dim=D, scale=scale, q_stride=C, k_stride=2 * C, v_stride=2 * C)
    y_chain = ops.linear(o, ops.pack_conv(wo, bo, device=dev), residual=x, out_f32=True)
    # the fused launch
    assert ops.xattn_ok(x, heads=H, head_dim=D, lk=lk, rows_per_kv=rpk)
    wqp, wop = ops.pack_xattn_weight(wq, "q", dev), ops.pack_xattn_weight(wo, "out", dev)
    kvp = ops.xattn_pack_kv(k, v, n_batch=nb, lk=lk, k_stride=2 * C, v_stride=2 * C)
    y = ops.xattn_sublayer(x, gamma, beta, 1e-5, wqp, kvp, wop, bo, rows_per_kv=rpk, lk=lk, scale=scale)
    assert y.dtype == torch.float32 and y.shape == x.shape and bool(torch.isfinite(y).all())
    e_out = rel_l2(y, y_chain)
    e_upd = rel_l2(y - x, y_chain - x)
    # (b) fp32 reference of the sub-layer
    nf = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    qh = (nf @ wq.t()).reshape(nb, rpk, H, D).permute(0, 2, 1, 3)
    kh = k.float().reshape(nb, lk, H, D).permute(0, 2, 1, 3); vh = v.float().reshape(nb, lk, H, D).permute(0, 2, 1, 3)
    att = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh
    ref = x + att.permute(0, 2, 1, 3).reshape(M, C) @ wo.t() + bo
    e_ref, e_ref_chain = rel_l2(y - x, ref - x), rel_l2(y_chain - x, ref - x)
    assert e_out < 2e-4, (name, e_out, e_upd)                 # whole rows (residual included) against the chain
    assert e_upd < 2e-3, (name, e_upd)                        # the update alone: fp16-rounding flips of Q / P / O only
    assert e_ref < 2.5e-3 and e_ref < 1.25 * e_ref_chain + 1e-4, (name, e_ref, e_ref_chain)      # as close to fp32 as the chain is
    # in place (out = x): a workgroup reads its rows before it writes them
    x2 = x.clone()
    y2 = ops.xattn_sublayer(x2, gamma, beta, 1e-5, wqp, kvp, wop, bo, rows_per_kv=rpk, lk=lk, scale=scale, out=x2)
    assert y2 is x2 and torch.equal(x2, y)
    # deterministic
    assert torch.equal(ops.xattn_sublayer(x, gamma, beta, 1e-5, wqp, kvp, wop, bo, rows_per_kv=rpk, lk=lk, scale=scale), y)


def test_fused_cross_attention_block_matches_four_launch_chain(ops, dev):
    """BasicTransformerBlock (only_cross_attention: attn1 AND attn2 are text cross-attention, attention.py:523-564) with the fused
    sub-layer kernel on and off: same module, same weights, fp32 token stream."""
    from uav import engine as E
    from models_video.attention import BasicTransformerBlock
    g = torch.Generator().manual_seed(77)
    blk = BasicTransformerBlock(512, 8, 64, cross_attention_dim=1024, only_cross_attention=True)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.2))
        for ln in (blk.norm1, blk.norm2, blk.norm_temporal, blk.norm3):
            ln.weight.add_(1.0)
    blk = blk.half().to(dev).eval()
    geom = E.Geom(2, 4, 16, 16)                                 # rows per kv batch 1024
    x = (torch.randn(geom.rows, 512, generator=g) * 1.2).to(dev)
    ehs = (torch.randn(2 * 77, 1024, generator=g)).half().to(dev)
    old = E.XATTN_FUSED
    try:
        E.XATTN_FUSED = True
        E.invalidate_packed(blk)
        with torch.no_grad():
            y1 = blk.run(x.clone(), geom, ehs, 77)
        E.XATTN_FUSED = False
        E.invalidate_packed(blk)
        with torch.no_grad():
            y0 = blk.run(x.clone(), geom, ehs, 77)
    finally:
        E.XATTN_FUSED = old
        E.invalidate_packed(blk)
    assert y1.dtype == y0.dtype == torch.float32
    e = rel_l2(y1, y0)
    # measured 6.1e-4 (round 6, run 1): with these unscaled random weights the block output is dominated by the sub-layers' updates,
    # and the two forms differ by fp16-rounding flips of Q / P / O on the update (<= 2e-3 of it, asserted per sub-layer above)
    assert e < 1.5e-3, e


# ------------------------------------------------------------------------------------------------
# Block tails as the hi | lo operand pair of their 1x1 consumer, written by the producer's epilogue (round 6, UAV_CONV_OUT_HILO):
# bit-identical to the fp32 result followed by uav_cast_f32_hilo — attention.py:389-398 (last feed-forward -> proj_out),
# temporal_module.py:175-194 (tail ResNet -> shift_conv)
HILO_CASES = [
    # name, cin, cout, k3, n_img, t_len, h, w, extras
    ("ff_down_2048_512_res32", 2048, 512, (1, 1, 1), 2, 1, 32768, 1, dict(res=True)),
    ("resnet_conv2_3x3_256_res32_scaled", 256, 256, (1, 3, 3), 4, 2, 128, 128, dict(res=True, scale=1 / 1.3)),
    ("no_residual_falls_back_to_cast_pass", 512, 512, (1, 1, 1), 1, 1, 61440, 1, dict(fallback=True)),
    ("small_grid_falls_back_to_cast_pass", 512, 512, (1, 1, 1), 1, 1, 4096, 1, dict(res=True, fallback=True)),
    ("row_tail_falls_back_to_cast_pass", 512, 512, (1, 1, 1), 1, 1, 61447, 1, dict(res=True, fallback=True)),
]


@pytest.mark.parametrize("case", HILO_CASES, ids=[c[0] for c in HILO_CASES])
def test_conv_result_as_hilo_pair_is_bit_identical_to_cast_pass(ops, dev, case):
    name, cin, cout, k3, n_img, t_len, h, w, e = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    rows = n_img * h * w
    x = torch.randn(rows, cin, generator=g).half().to(dev)
    wt = h16(cout, cin, *k3, dev=dev, scale=(cin * k3[1] * k3[2]) ** -0.5, gen=g)
    cw = ops.pack_conv(wt, torch.randn(cout, generator=g).to(dev), device=dev)
    res = (torch.randn(rows, cout, generator=g) * 3).to(dev) if e.get("res") else None
    kw = dict(n_img=n_img, t_len=t_len, hi=h, wi=w, residual=res, out_scale=e.get("scale", 1.0), out_f32=True)
    y32 = ops.conv_gemm(x, cw, **kw)
    pair = ops.conv_gemm(x, cw, out_hilo=True, **kw)
    assert pair.dtype == torch.float16 and pair.shape == (rows, 2 * cout)
    assert torch.equal(pair, ops.cast_hilo(y32)), name
    # the pair carries the fp32 value to ~2^-22: hi + lo against the fp32 result
    back = pair[:, :cout].float() + pair[:, cout:].float()
    assert rel_l2(back, y32) < 2e-6
    # which path stored it
    from uav import _lib
    import ctypes as C
    lib = _lib.load()
    p = _lib.ConvParams()
    p.a1 = x.data_ptr(); p.c1 = cin; p.w = cw.w.data_ptr(); p.bias = cw.bias.data_ptr(); p.n_img = n_img; p.t_len = t_len; p.hi = h; p.wi = w
    p.ho = h; p.wo = w; p.kt, p.kh, p.kw = cw.kt, cw.kh, cw.kw; p.stride = 1; p.pad_h = k3[1] // 2; p.pad_w = k3[2] // 2
    p.n = cw.n; p.n_pad = cw.n_pad; p.k_pad = cw.k_pad; p.out = pair.data_ptr(); p.out_stride = 2 * cout
    p.flags = _lib.CONV_OUT_F32 | _lib.CONV_OUT_HILO | (_lib.CONV_RES_F32 if res is not None else 0)
    if res is not None:
        p.residual = res.data_ptr(); p.res_stride = cout
    assert bool(lib.uav_conv_gemm_hilo_ok(C.byref(p))) == (not e.get("fallback", False))


def test_fused_cross_attention_pair_equals_two_single_launches(ops, dev):
    """n_subs = 2 (attn1 with only_cross_attention + attn2 of one block, attention.py:523-564): the second LayerNorm runs on the accumulators
    that hold the first sub-layer's fp32 output — the same values a first launch would have stored and a second one re-read, so the pair
    must reproduce two single launches up to the fp32 summation order of the LayerNorm statistics."""
    g = torch.Generator().manual_seed(321)
    C, D, nb, rpk, lk = 512, 64, 2, 1280, 77
    M = nb * rpk
    x = (torch.randn(M, C, generator=g) * 1.4 + 0.5).to(dev)
    subs = []
    for _ in range(2):
        gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
        wq = h16(C, C, dev=dev, scale=C ** -0.5, gen=g); wo = h16(C, C, dev=dev, scale=C ** -0.5, gen=g)
        bo = (torch.randn(C, generator=g) * 0.1).to(dev)
        kv = (torch.randn(nb * lk, 2 * C, generator=g) * 1.5).half().to(dev)
        kvp = ops.xattn_pack_kv(kv[:, :C], kv[:, C:], n_batch=nb, lk=lk, k_stride=2 * C, v_stride=2 * C)
        subs.append((gamma, beta, 1e-5, ops.pack_xattn_weight(wq, "q", dev), kvp, ops.pack_xattn_weight(wo, "out", dev), bo))
    kw = dict(rows_per_kv=rpk, lk=lk, scale=D ** -0.5)
    y1 = ops.xattn_sublayers(x, [subs[0]], **kw)
    y2 = ops.xattn_sublayers(y1, [subs[1]], **kw)
    yp = ops.xattn_sublayers(x, subs, **kw)
    assert bool(torch.isfinite(yp).all())
    e = rel_l2(yp, y2)
    e_upd = rel_l2(yp - y1, y2 - y1)
    assert e < 5e-5 and e_upd < 1e-3, (e, e_upd)              # LayerNorm statistics: two-pass in registers vs the shifted one-pass of the first read
    assert torch.equal(ops.xattn_sublayers(x, subs, **kw), yp)
    xin = x.clone()
    assert ops.xattn_sublayers(xin, subs, out=xin, **kw) is xin and torch.equal(xin, yp)      # in place


# ------------------------------------------------------------------------------------------------
# Fused TEMPORAL attention sub-layer (round 6, tattn_sublayer_kernel): LayerNorm -> to_q | to_k | to_v -> RoPE + relative-position bias +
# softmax over the 8 frames of a pixel -> to_out -> + residual in one launch — reference attention.py:555-560, TemporalAttention :626-733
TATTN_CASES = [
    # name, batches, h, w, row mean / spread
    ("b2_16x16", 2, 16, 16, 0.3, 1.5),
    ("b1_4x4_one_tile", 1, 4, 4, 0.0, 1.0),
    ("b2_40x40", 2, 40, 40, -1.0, 0.8),
    ("b1_160x160_level", 1, 160, 160, 2.0, 2.0),
]


@pytest.mark.parametrize("case", TATTN_CASES, ids=[c[0] for c in TATTN_CASES])
def test_fused_temporal_attention_sublayer(ops, dev, case):
    name, nb, hh, ww, mu, sd = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    C, H, D, T = 512, 8, 64, 8
    hw = hh * ww
    M = nb * T * hw
    x = (torch.randn(M, C, generator=g) * sd + mu + torch.randn(M, 1, generator=g)).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    wq, wk, wv, wo = (h16(C, C, dev=dev, scale=C ** -0.5, gen=g) for _ in range(4))
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    relb = (torch.randn(H, T, T, generator=g) * 0.5).to(dev).contiguous()
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(T).float()[:, None] * fr[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    scale = D ** -0.5
    # the four-launch chain
    n = ops.layernorm(x, gamma, beta, 1e-5)
    qkv = ops.linear(n, ops.pack_conv(torch.cat([wq, wk, wv], 0), None, device=dev))
    o = ops.temporal_attention(qkv, n_batch=nb, t_len=T, hw=hw, c=C, heads=H, scale=scale, rope_cos=cos, rope_sin=sin, rot_dim=32, bias=relb)
    y_chain = ops.linear(o, ops.pack_conv(wo, bo, device=dev), residual=x, out_f32=True)
    # the fused launch
    assert ops.tattn_ok(x, heads=H, head_dim=D, t_len=T, hw=hw, rot_dim=32)
    pk = [ops.pack_xattn_weight(w_, "q", dev) for w_ in (wq, wk, wv)] + [ops.pack_xattn_weight(wo, "out", dev)]
    kw = dict(n_batch=nb, t_len=T, hw=hw, rot_dim=32, scale=scale)
    y = ops.tattn_sublayer(x, gamma, beta, 1e-5, *pk, bo, relb, cos, sin, **kw)
    assert y.dtype == torch.float32 and y.shape == x.shape and bool(torch.isfinite(y).all())
    # fp32 reference (TemporalAttention.forward on rows (b, t, p))
    nf = F.layer_norm(x, (C,), gamma, beta, 1e-5)

    def heads_of(t):
        return t.reshape(nb, T, hw, H, D).permute(0, 2, 3, 1, 4)              # (B, P, H, T, d)

    def rope(t):
        c_, s_ = cos.reshape(1, 1, 1, T, 16), sin.reshape(1, 1, 1, T, 16)
        a, b = t[..., 0:32:2], t[..., 1:32:2]
        r = torch.stack([a * c_ - b * s_, b * c_ + a * s_], dim=-1).reshape(t.shape[:-1] + (32,))
        return torch.cat([r, t[..., 32:]], dim=-1)
    qh, kh, vh = rope(heads_of(nf @ wq.t()) * scale), rope(heads_of(nf @ wk.t())), heads_of(nf @ wv.t())
    att = torch.softmax(qh @ kh.transpose(-1, -2) + relb.reshape(1, 1, H, T, T), dim=-1) @ vh      # (B, P, H, T, d)
    ref = x + att.permute(0, 3, 1, 2, 4).reshape(M, C) @ wo.t() + bo
    e_out, e_upd = rel_l2(y, y_chain), rel_l2(y - x, y_chain - x)
    e_ref, e_ref_chain = rel_l2(y - x, ref - x), rel_l2(y_chain - x, ref - x)
    assert e_out < 2e-4, (name, e_out, e_upd)
    assert e_upd < 2.5e-3, (name, e_upd)
    assert e_ref < 2.5e-3 and e_ref < 1.3 * e_ref_chain + 1e-4, (name, e_ref, e_ref_chain)
    x2 = x.clone()
    assert ops.tattn_sublayer(x2, gamma, beta, 1e-5, *pk, bo, relb, cos, sin, out=x2, **kw) is x2 and torch.equal(x2, y)     # in place
    assert torch.equal(ops.tattn_sublayer(x, gamma, beta, 1e-5, *pk, bo, relb, cos, sin, **kw), y)                             # deterministic


def test_fused_block_attention_three_sublayers_equal_pair_then_temporal(ops, dev):
    """uav_block_attn_sublayers_f32 (attn1 -> attn2 -> attn_temporal of a block with only_cross_attention in one launch, attention.py:523-564)
    against the cross-attention pair launch followed by the temporal launch on the same rows."""
    g = torch.Generator().manual_seed(99)
    C, H, D, T, nb, hh, ww, lk = 512, 8, 64, 8, 2, 24, 16, 77
    hw = hh * ww
    M = nb * T * hw
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.4).to(dev)
    cross = []
    for _ in range(2):
        gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
        wq = h16(C, C, dev=dev, scale=C ** -0.5, gen=g); wo = h16(C, C, dev=dev, scale=C ** -0.5, gen=g)
        bo = (torch.randn(C, generator=g) * 0.1).to(dev)
        kv = (torch.randn(nb * lk, 2 * C, generator=g) * 1.5).half().to(dev)
        kvp = ops.xattn_pack_kv(kv[:, :C], kv[:, C:], n_batch=nb, lk=lk, k_stride=2 * C, v_stride=2 * C)
        cross.append((gamma, beta, 1e-5, ops.pack_xattn_weight(wq, "q", dev), kvp, ops.pack_xattn_weight(wo, "out", dev), bo))
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    ws = [h16(C, C, dev=dev, scale=C ** -0.5, gen=g) for _ in range(4)]
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    relb = (torch.randn(H, T, T, generator=g) * 0.5).to(dev).contiguous()
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(T).float()[:, None] * fr[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    temporal = (gamma, beta, 1e-5, *[ops.pack_xattn_weight(w_, "q", dev) for w_ in ws[:3]], ops.pack_xattn_weight(ws[3], "out", dev), bo, relb, cos, sin, 32)
    scale = D ** -0.5
    y1 = ops.xattn_sublayers(x, cross, rows_per_kv=T * hw, lk=lk, scale=scale)
    y2 = ops.tattn_sublayer(y1, *temporal[:11], n_batch=nb, t_len=T, hw=hw, rot_dim=32, scale=scale)
    kw = dict(n_batch=nb, t_len=T, hw=hw, lk=lk, cross_scale=scale, temporal_scale=scale)
    y3 = ops.block_attn_sublayers(x, cross, temporal, **kw)
    assert bool(torch.isfinite(y3).all())
    e, e_upd = rel_l2(y3, y2), rel_l2(y3 - x, y2 - x)
    assert e < 1e-4 and e_upd < 1.5e-3, (e, e_upd)            # LayerNorm statistics from the accumulators vs from the shifted one-pass first read
    assert torch.equal(ops.block_attn_sublayers(x, cross, temporal, **kw), y3)
    xin = x.clone()
    assert ops.block_attn_sublayers(xin, cross, temporal, out=xin, **kw) is xin and torch.equal(xin, y3)


def test_fused_sublayer_kernels_emit_the_next_layernorm(ops, dev):
    """`next_ln`: the temporal / block kernels also write fp16 LayerNorm(y) of the rows they store (the block's norm3, attention.py:562-564), so
    the feed-forward's LayerNorm launch disappears — against the LayerNorm kernel on the same rows."""
    g = torch.Generator().manual_seed(4242)
    C, H, D, T, nb, hh, ww = 512, 8, 64, 8, 2, 16, 16
    hw = hh * ww
    M = nb * T * hw
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.4).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    ws = [h16(C, C, dev=dev, scale=C ** -0.5, gen=g) for _ in range(4)]
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    relb = (torch.randn(H, T, T, generator=g) * 0.5).to(dev).contiguous()
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(T).float()[:, None] * fr[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    pk = [ops.pack_xattn_weight(w_, "q", dev) for w_ in ws[:3]] + [ops.pack_xattn_weight(ws[3], "out", dev)]
    g3 = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); b3 = (torch.randn(C, generator=g) * 0.1).to(dev)
    kw = dict(n_batch=nb, t_len=T, hw=hw, rot_dim=32, scale=D ** -0.5)
    y0 = ops.tattn_sublayer(x, gamma, beta, 1e-5, *pk, bo, relb, cos, sin, **kw)
    y = ops.tattn_sublayer(x, gamma, beta, 1e-5, *pk, bo, relb, cos, sin, next_ln=(g3, b3, 1e-5), **kw)
    assert torch.equal(y, y0)                                        # the fp32 rows themselves are unchanged
    n = ops.next_ln_of(y, g3, b3, 1e-5)
    assert n is not None and n.dtype == torch.float16 and n.shape == y.shape
    ref = ops.layernorm(y, g3, b3, 1e-5)
    assert rel_l2(n, ref) < 1e-4 and (n.float() - ref.float()).abs().max().item() < 4e-3      # same two-pass arithmetic, fp32 summation order apart
    assert ops.next_ln_of(y, b3, g3, 1e-5) is None and ops.next_ln_of(y, g3, b3, 1e-6) is None   # other parameters: not this LayerNorm
    y.add_(1.0)
    assert ops.next_ln_of(y, g3, b3, 1e-5) is None                   # rows changed since: stale


@pytest.mark.parametrize("m,mode", [(128, "f32"), (4096, "f32"), (2048, "hilo"), (1024, "both")])
def test_fused_feed_forward_sublayer(ops, dev, m, mode):
    """uav_ff_sublayer_f32 (reference attention.py:562-564 `ff(norm3(x)) + x`, GEGLU feed-forward of diffusers) against the three launches it
    replaces (LayerNorm, 512 -> 4096 GEMM with the GEGLU epilogue, 2048 -> 512 GEMM + residual) — same roundings, fp32 summation order
    apart — and against fp32 torch; the hi | lo pair is cast_hilo of the fp32 rows, bit for bit."""
    g = torch.Generator().manual_seed(77 + m)
    C, I = 512, 2048
    x = (torch.randn(m, C, generator=g) * 1.3 + 0.4).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    wu = h16(2 * I, C, dev=dev, scale=C ** -0.5, gen=g); wd = h16(C, I, dev=dev, scale=I ** -0.5, gen=g)
    bu = (torch.randn(2 * I, generator=g) * 0.2).to(dev); bd = (torch.randn(C, generator=g) * 0.1).to(dev)
    wp = ops.pack_ff_weights(wu, wd, dev)
    assert ops.ff_ok(x, inner=I)
    r = ops.ff_sublayer(x, gamma, beta, 1e-5, wp, bu, bd, out_f32=mode != "hilo", out_hilo=mode != "f32")
    y, yh = (r, None) if mode == "f32" else (None, r) if mode == "hilo" else r
    # the chain
    n = ops.layernorm(x, gamma, beta, 1e-5)
    hdn = ops.linear(n, ops.pack_conv(wu, bu, geglu=True, device=dev))
    yc = ops.linear(hdn, ops.pack_conv(wd, bd, device=dev), residual=x, out_f32=True)
    # fp32 torch on the same fp16-representable weights
    nf = torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5)
    u = nf @ wu.float().t() + bu
    yf = x + bd + (u[:, :I] * torch.nn.functional.gelu(u[:, I:])) @ wd.float().t()
    if y is not None:
        assert rel_l2(y - x, yc - x) < 3e-4, rel_l2(y - x, yc - x)      # the branch alone (the residual would hide it)
        assert rel_l2(y - x, yf - x) < 2e-3
        assert torch.isfinite(y).all()
    if yh is not None:
        assert yh.shape == (m, 2 * C) and yh.dtype == torch.float16
        if y is not None:
            assert torch.equal(yh, ops.cast_hilo(y))
        else:
            yy = yh[:, :C].float() + yh[:, C:].float()
            assert rel_l2(yy - x, yc - x) < 3e-4
    # determinism
    r2 = ops.ff_sublayer(x, gamma, beta, 1e-5, wp, bu, bd, out_f32=mode != "hilo", out_hilo=mode != "f32")
    assert all(torch.equal(a, b) for a, b in zip(r if isinstance(r, tuple) else (r,), r2 if isinstance(r2, tuple) else (r2,)))
    with pytest.raises(Exception):
        ops.ff_sublayer(x[:100], gamma, beta, 1e-5, wp, bu, bd)           # not a whole 128-row tile


def test_transformer_block_with_fused_feed_forward_matches_three_launch_chain(ops, dev):
    """BasicTransformerBlock (T = 8: the block kernel in front) with the feed-forward as one launch on and off — same module, same weights —,
    as fp32 rows and as the hi | lo pair of proj_out."""
    from uav import engine as E
    from models_video.attention import BasicTransformerBlock
    g = torch.Generator().manual_seed(78)
    blk = BasicTransformerBlock(512, 8, 64, cross_attention_dim=1024, only_cross_attention=True)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.2))
        for ln in (blk.norm1, blk.norm2, blk.norm_temporal, blk.norm3):
            ln.weight.add_(1.0)
    blk = blk.half().to(dev).eval()
    geom = E.Geom(2, 8, 16, 16)
    x = (torch.randn(geom.rows, 512, generator=g) * 1.2).to(dev)
    ehs = (torch.randn(2 * 77, 1024, generator=g)).half().to(dev)
    old = E.FF_FUSED
    res = {}
    try:
        for on in (True, False):
            E.FF_FUSED = on
            E.invalidate_packed(blk)
            with torch.no_grad():
                res[on] = (blk.run(x.clone(), geom, ehs, 77), blk.run(x.clone(), geom, ehs, 77, out_hilo=True))
    finally:
        E.FF_FUSED = old
        E.invalidate_packed(blk)
    (y1, h1), (y0, h0) = res[True], res[False]
    assert y1.dtype == y0.dtype == torch.float32 and h1.dtype == h0.dtype == torch.float16
    e = rel_l2(y1, y0)
    assert e < 5e-4, e
    assert torch.equal(h1, ops.cast_hilo(y1)) and torch.equal(h0, ops.cast_hilo(y0))


@pytest.mark.parametrize("mode", ["f32", "hilo", "both"])
def test_whole_transformer_block_in_one_launch_equals_attention_launch_then_feed_forward_launch(ops, dev, mode):
    """uav_block_sublayers_f32 (attn1 -> attn2 -> attn_temporal -> ff of a BasicTransformerBlock in one launch, attention.py:523-564) against
    the three-attention launch followed by the feed-forward launch on the same rows; the hi | lo pair is cast_hilo of the fp32 rows."""
    g = torch.Generator().manual_seed(123)
    C, H, D, T, nb, hh, ww, lk, I = 512, 8, 64, 8, 2, 24, 16, 77, 2048
    hw = hh * ww
    M = nb * T * hw
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.4).to(dev)
    cross = []
    for _ in range(2):
        gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
        wq = h16(C, C, dev=dev, scale=C ** -0.5, gen=g); wo = h16(C, C, dev=dev, scale=C ** -0.5, gen=g)
        bo = (torch.randn(C, generator=g) * 0.1).to(dev)
        kv = (torch.randn(nb * lk, 2 * C, generator=g) * 1.5).half().to(dev)
        kvp = ops.xattn_pack_kv(kv[:, :C], kv[:, C:], n_batch=nb, lk=lk, k_stride=2 * C, v_stride=2 * C)
        cross.append((gamma, beta, 1e-5, ops.pack_xattn_weight(wq, "q", dev), kvp, ops.pack_xattn_weight(wo, "out", dev), bo))
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    ws = [h16(C, C, dev=dev, scale=C ** -0.5, gen=g) for _ in range(4)]
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    relb = (torch.randn(H, T, T, generator=g) * 0.5).to(dev).contiguous()
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(T).float()[:, None] * fr[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    temporal = (gamma, beta, 1e-5, *[ops.pack_xattn_weight(w_, "q", dev) for w_ in ws[:3]], ops.pack_xattn_weight(ws[3], "out", dev), bo, relb, cos, sin, 32)
    g3 = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); b3 = (torch.randn(C, generator=g) * 0.1).to(dev)
    wu = h16(2 * I, C, dev=dev, scale=C ** -0.5, gen=g); wd = h16(C, I, dev=dev, scale=I ** -0.5, gen=g)
    bu = (torch.randn(2 * I, generator=g) * 0.2).to(dev); bd = (torch.randn(C, generator=g) * 0.1).to(dev)
    ff = (g3, b3, 1e-5, ops.pack_ff_weights(wu, wd, dev), bu, bd)
    scale = D ** -0.5
    kw = dict(n_batch=nb, t_len=T, hw=hw, lk=lk, cross_scale=scale, temporal_scale=scale)
    y3 = ops.block_attn_sublayers(x, cross, temporal, **kw)
    y4 = ops.ff_sublayer(y3, *ff)
    r = ops.block_sublayers(x, cross, temporal, ff, out_f32=mode != "hilo", out_hilo=mode != "f32", **kw)
    y, yh = (r, None) if mode == "f32" else (None, r) if mode == "hilo" else r
    if y is not None:
        assert bool(torch.isfinite(y).all())
        e, e_upd = rel_l2(y, y4), rel_l2(y - x, y4 - x)
        assert e < 1e-4 and e_upd < 1.5e-3, (e, e_upd)        # the feed-forward's LayerNorm statistics from the accumulators vs from its own first read
    if yh is not None:
        assert yh.shape == (M, 2 * C) and yh.dtype == torch.float16
        if y is not None:
            assert torch.equal(yh, ops.cast_hilo(y))
        else:
            assert rel_l2(yh[:, :C].float() + yh[:, C:].float(), y4) < 1e-4
    r2 = ops.block_sublayers(x, cross, temporal, ff, out_f32=mode != "hilo", out_hilo=mode != "f32", **kw)
    assert all(torch.equal(a, b) for a, b in zip(r if isinstance(r, tuple) else (r,), r2 if isinstance(r2, tuple) else (r2,)))


def test_whole_block_launch_with_groupnorm_apply_and_proj_in_in_front(ops, dev):
    """uav_block_sublayers_f32 with proj_in: GroupNorm apply -> proj_in -> attn1 -> attn2 -> attn_temporal -> ff in one launch (Transformer3DModel,
    attention.py:389-398 up to proj_out) against GroupNorm (scale / shift + apply pass), the proj_in GEMM launch and the whole-block launch."""
    g = torch.Generator().manual_seed(321)
    C, H, D, T, nb, hh, ww, lk, I = 512, 8, 64, 8, 2, 16, 16, 77, 2048
    hw = hh * ww
    M = nb * T * hw
    x = (torch.randn(M, C, generator=g) * 1.7 + 0.6).to(dev)
    gg = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); gb = (torch.randn(C, generator=g) * 0.1).to(dev)
    w_in = h16(C, C, dev=dev, scale=C ** -0.5, gen=g); b_in = (torch.randn(C, generator=g) * 0.1).to(dev)
    cross = []
    for _ in range(2):
        gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
        wq = h16(C, C, dev=dev, scale=C ** -0.5, gen=g); wo = h16(C, C, dev=dev, scale=C ** -0.5, gen=g)
        bo = (torch.randn(C, generator=g) * 0.1).to(dev)
        kv = (torch.randn(nb * lk, 2 * C, generator=g) * 1.5).half().to(dev)
        kvp = ops.xattn_pack_kv(kv[:, :C], kv[:, C:], n_batch=nb, lk=lk, k_stride=2 * C, v_stride=2 * C)
        cross.append((gamma, beta, 1e-5, ops.pack_xattn_weight(wq, "q", dev), kvp, ops.pack_xattn_weight(wo, "out", dev), bo))
    gamma = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    ws = [h16(C, C, dev=dev, scale=C ** -0.5, gen=g) for _ in range(4)]
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    relb = (torch.randn(H, T, T, generator=g) * 0.5).to(dev).contiguous()
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(T).float()[:, None] * fr[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    temporal = (gamma, beta, 1e-5, *[ops.pack_xattn_weight(w_, "q", dev) for w_ in ws[:3]], ops.pack_xattn_weight(ws[3], "out", dev), bo, relb, cos, sin, 32)
    g3 = (torch.randn(C, generator=g) * 0.2 + 1.0).to(dev); b3 = (torch.randn(C, generator=g) * 0.1).to(dev)
    wu = h16(2 * I, C, dev=dev, scale=C ** -0.5, gen=g); wd = h16(C, I, dev=dev, scale=I ** -0.5, gen=g)
    bu = (torch.randn(2 * I, generator=g) * 0.2).to(dev); bd = (torch.randn(C, generator=g) * 0.1).to(dev)
    ff = (g3, b3, 1e-5, ops.pack_ff_weights(wu, wd, dev), bu, bd)
    scale = D ** -0.5
    kw = dict(n_batch=nb, t_len=T, hw=hw, lk=lk, cross_scale=scale, temporal_scale=scale)
    # the three-launch form
    sc, sh = ops.groupnorm_scale_shift(x, gg, gb, n_inst=nb * T, rows_per_inst=hw, groups=32, eps=1e-6)
    n = ops.groupnorm_apply(x, sc, sh, n_inst=nb * T, rows_per_inst=hw, silu=False)
    tok = ops.linear(n, ops.pack_conv(w_in, b_in, device=dev), out_f32=True)
    y_ref, h_ref = ops.block_sublayers(tok, cross, temporal, ff, out_f32=True, out_hilo=True, **kw)
    # one launch
    pi = (sc, sh, ops.pack_xattn_weight(w_in, "out", dev), b_in)
    y, yh = ops.block_sublayers(x, cross, temporal, ff, out_f32=True, out_hilo=True, proj_in=pi, **kw)
    assert bool(torch.isfinite(y).all())
    e, e_upd = rel_l2(y, y_ref), rel_l2(y - tok, y_ref - tok)
    # tok agrees to fp32 summation order, but norm1's statistics now come from the accumulators (two passes) instead of the shifted one-pass
    # first read: fp16 rounding flips of the LayerNorm rows in ALL four sub-layers (measured 3.3e-4 / 4.5e-4 with these unscaled random
    # weights, whose updates are as large as the stream; a wrong table, bias or frame mapping would show as 1e-1)
    assert e < 6e-4 and e_upd < 1.5e-3, (e, e_upd)
    assert torch.equal(yh, ops.cast_hilo(y))
    yh2 = ops.block_sublayers(x, cross, temporal, ff, out_f32=False, out_hilo=True, proj_in=pi, **kw)
    assert torch.equal(yh2, yh)


def test_attention_d512_ring_kernel_on_fused_qkv_rows_and_against_the_old_kernel(ops, dev):
    """uav_attention512_pack_kv + uav_attention512_packed_f16 (VAE mid-block attention, vae.py AttentionBlock) on q | k | v slices of one fused
    projection (row stride 1536) against fp32 torch and against attn512w_kernel (uav_attention_f16) on the same rows."""
    from uav import _lib
    g = torch.Generator().manual_seed(512)
    bq, L, d = 2, 1500, 512
    qkv = (torch.randn(bq * L, 3 * d, generator=g) * 0.9).half().to(dev)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    out = ops.attention(q, k, v, bq=bq, lq=L, lk=L, heads=1, head_dim=d, q_stride=3 * d, k_stride=3 * d, v_stride=3 * d)
    qf, kf, vf = (t.float().reshape(bq, L, d) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5, dim=-1) @ vf).reshape(bq * L, d)
    assert rel_l2(out, ref) < 3e-3
    lib = _lib.load()
    old = torch.empty_like(out)
    rc = lib.uav_attention_f16(q.data_ptr(), 3 * d, k.data_ptr(), 3 * d, v.data_ptr(), 3 * d, old.data_ptr(), d, bq, L, L, 1, 1, d, d ** -0.5, 0,
                               ops.zero_page(dev).data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert rel_l2(out, old) < 1.5e-3                          # both round P to fp16; another key-tile size and summation order
    assert torch.equal(ops.attention(q, k, v, bq=bq, lq=L, lk=L, heads=1, head_dim=d, q_stride=3 * d, k_stride=3 * d, v_stride=3 * d), out)
