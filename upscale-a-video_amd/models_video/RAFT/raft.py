"""MI355X-native RAFT ("things" configuration) — drop-in for the reference's
`models_video/RAFT/raft.py` (RAFT.forward :87-145), `extractor.py` (BasicEncoder :118-193,
ResidualBlock :6-55), `update.py` (BasicUpdateBlock :114-139, BasicMotionEncoder :82-103, SepConvGRU
:33-60, FlowHead :6-14) and `corr.py` (CorrBlock :12-61).  Same module tree / state-dict keys as the
reference (tests/golden/raft_keys.json), fp32 like the reference (raft_bi.py:26), all compute in
libuav_hip.so on channels-last fp32 rows:

  every conv (7x7/s2, 3x3, 1x1, 1x5, 5x1)   uav_conv_gemm_f32 (exact-fp32 MFMA) with bias + ReLU /
                                            sigmoid / tanh epilogues; eval-mode BatchNorm of the
                                            context encoder is folded into the conv weights
  InstanceNorm (+ReLU)                      uav_instnorm_f32
  all-pairs correlation                     uav_conv_gemm_f32 as a linear whose weight is fmap2
  pyramid / 9x9x4 lookup / convex upsample  uav_avgpool2_f32 / uav_corr_lookup_f32 / uav_convex_upsample_f32
  GRU gates                                 convz|convr fused into one conv, uav_gru_gates_f32

Work the reference repeats and this engine does once (identical values): the feature / context
encoders see every FRAME once instead of every frame of every (image1, image2) pair in both
directions (4(T-1) -> T encoder passes); both flow directions are batched through the GRU loop.
"""
import torch
import torch.nn as nn

from uav import engine as E
from uav import ops


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.stride, self.norm_fn = stride, norm_fn
        mk = {"batch": lambda: nn.BatchNorm2d(planes), "instance": lambda: nn.InstanceNorm2d(planes),
              "none": lambda: nn.Sequential()}[norm_fn]
        self.norm1, self.norm2 = mk(), mk()
        if stride == 1:
            self.downsample = None
        else:
            self.norm3 = mk()
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)


class BasicEncoder(E.EngineModule):
    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        if norm_fn not in ("batch", "instance"):
            raise NotImplementedError(norm_fn)
        self.norm_fn = norm_fn
        self.norm1 = nn.BatchNorm2d(64) if norm_fn == "batch" else nn.InstanceNorm2d(64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, norm_fn, 1), ResidualBlock(64, 64, norm_fn, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, norm_fn, 2), ResidualBlock(96, 96, norm_fn, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, norm_fn, 2), ResidualBlock(128, 128, norm_fn, 1))
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)

    def _pack(self, key, conv, bn=None):
        """conv (+ folded eval-mode BatchNorm) -> packed fp32 weights."""
        def build():
            dev = E._dev(conv.weight)
            w, b = conv.weight.detach().float(), conv.bias.detach().float()
            if isinstance(bn, nn.BatchNorm2d):
                s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                w = w * s[:, None, None, None]
                b = (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
            return ops.pack_conv_f32(w, b, device=dev)
        return self._cache().get(("c", key), build, (conv.weight, conv.bias) + ((bn.weight, bn.bias, bn.running_mean, bn.running_var)
                                                                           if isinstance(bn, nn.BatchNorm2d) else ()))

    def _conv_norm(self, key, x, conv, norm, n, h, w, relu):
        s, p = conv.stride[0], conv.padding
        ho, wo = (h + 2 * p[0] - conv.kernel_size[0]) // s + 1, (w + 2 * p[1] - conv.kernel_size[1]) // s + 1
        if self.norm_fn == "batch":
            y = ops.conv_gemm_f32(x, self._pack(key, conv, norm), n_img=n, hi=h, wi=w, stride=s, pad=tuple(p), act=1 if relu else 0)
        else:
            y = ops.conv_gemm_f32(x, self._pack(key, conv), n_img=n, hi=h, wi=w, stride=s, pad=tuple(p))
            y = ops.instnorm_f32(y, n_img=n, hw=ho * wo, relu=relu)
        return y, ho, wo

    def run(self, x, n, h, w):
        """x: rows [n*h*w][4] (3 real channels) -> rows [n*(h/8)*(w/8)][output_dim]."""
        x, h, w = self._conv_norm("conv1", x, self.conv1, self.norm1, n, h, w, True)
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3)):
            for bi, blk in enumerate(layer):
                key = f"l{li}.{bi}"
                y, h2, w2 = self._conv_norm(key + ".c1", x, blk.conv1, blk.norm1, n, h, w, True)
                y, _, _ = self._conv_norm(key + ".c2", y, blk.conv2, blk.norm2, n, h2, w2, True)
                if blk.downsample is not None:
                    x, _, _ = self._conv_norm(key + ".ds", x, blk.downsample[0], blk.downsample[1], n, h, w, False)
                x = ops.add_relu_f32(x, y, True)
                h, w = h2, w2
        return ops.conv_gemm_f32(x, self._pack("conv2", self.conv2), n_img=n, hi=h, wi=w), h, w


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for sfx, k, p in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for g in "zrq":
                setattr(self, f"conv{g}{sfx}", nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p))


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)


class BasicUpdateBlock(E.EngineModule):
    def __init__(self, hidden_dim=128, input_dim=128):
        super().__init__()
        self.encoder = BasicMotionEncoder()
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1, padding=0))

    def pk(self, key, conv, cin_pad_to=None):
        return self._cache().get(("c", key), lambda: ops.pack_conv_f32(conv.weight, conv.bias, device=E._dev(conv.weight),
                                                                      cin_pad_to=cin_pad_to), (conv.weight, conv.bias))

    def pk_zr(self, sfx):
        def build():
            z, r = getattr(self.gru, "convz" + sfx), getattr(self.gru, "convr" + sfx)
            return ops.pack_conv_f32(torch.cat([z.weight.detach(), r.weight.detach()], 0),
                                     torch.cat([z.bias.detach(), r.bias.detach()], 0), device=E._dev(z.weight))
        z, r = getattr(self.gru, "convz" + sfx), getattr(self.gru, "convr" + sfx)
        return self._cache().get(("zr", sfx), build, (z.weight, z.bias, r.weight, r.bias))


class RAFT(E.EngineModule):
    CORR_PAD = 352                                           # 4*81 = 324 lookup channels padded to a multiple of 32

    def __init__(self, args=None):
        super().__init__()
        self.hidden_dim = self.context_dim = 128
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance")
        self.cnet = BasicEncoder(output_dim=256, norm_fn="batch")
        self.update_block = BasicUpdateBlock(hidden_dim=128)

    @torch.no_grad()
    def flows_bidirectional(self, frames, iters=20):
        """frames (T,3,H,W) fp32 on the GPU, H,W multiples of 8 -> (fwd, bwd) flows (T-1,2,H,W):
        RAFT.forward(test_mode=True) (reference raft.py:87-145) for the pairs (i -> i+1) and (i+1 -> i)."""
        t, c, H, W = frames.shape
        if H % 8 or W % 8 or t < 2:
            raise NotImplementedError("RAFT needs T >= 2 and H, W multiples of 8 (the reference pre-resizes otherwise)")
        dev = frames.device
        rows = torch.zeros((t * H * W, 4), dtype=torch.float32, device=dev)
        rows[:, :3] = frames.float().permute(0, 2, 3, 1).reshape(-1, 3)
        f, h, w = self.fnet.run(rows, t, H, W)                # [t*hw][256]
        cx, _, _ = self.cnet.run(rows, t, H, W)
        hw = h * w
        if h < 16 or w < 16:
            raise NotImplementedError("RAFT's 4-level correlation pyramid needs H, W >= 128")
        i1 = list(range(t - 1)) + list(range(1, t))           # image1 frame of each pair (fwd then bwd)
        i2 = list(range(1, t)) + list(range(t - 1))
        P = len(i1)
        M = P * hw
        # ---- all-pairs correlation volume + pyramid (corr.py:12-27,53-60) ------------------------
        hw4, npad = (hw + 3) // 4 * 4, (hw + 127) // 128 * 128
        fpad = torch.zeros((t, npad, 256), dtype=torch.float32, device=dev)
        fpad[:, :hw] = f.reshape(t, hw, 256)
        corr0 = torch.zeros((P, hw, hw4), dtype=torch.float32, device=dev)
        for p in range(P):
            wt = ops.ConvW(fpad[i2[p]], None, hw4, npad, 256, 256, 256, 1, 1, 1, False)
            ops.conv_gemm_f32(f[i1[p] * hw:(i1[p] + 1) * hw], wt, n_img=1, hi=hw, wi=1, out_scale=1.0 / 16.0, out=corr0[p])
        levels, strides, hs, ws = [corr0], [hw4], [h], [w]
        for _ in range(3):
            nxt = ops.avgpool2_f32(levels[-1], strides[-1], hs[-1], ws[-1], M)
            levels.append(nxt); strides.append(nxt.shape[1]); hs.append(hs[-1] // 2); ws.append(ws[-1] // 2)
        # ---- hidden state / context (raft.py:110-114) ------------------------------------------
        net = torch.empty((M, 128), dtype=torch.float32, device=dev)
        X = torch.zeros((M, 256), dtype=torch.float32, device=dev)           # cat[inp | motion(126) | flow(2)]
        for p in range(P):
            src = cx[i1[p] * hw:(i1[p] + 1) * hw]
            ops.copy_cols_f32(src, 0, net[p * hw:(p + 1) * hw], 0, 128, act=4)       # tanh
            ops.copy_cols_f32(src, 128, X[p * hw:(p + 1) * hw], 0, 128, act=1)       # relu
        ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        c0 = torch.zeros((hw, 4), dtype=torch.float32, device=dev)
        c0[:, 0] = xs.reshape(-1).float(); c0[:, 1] = ys.reshape(-1).float()
        coords0 = c0.repeat(P, 1).contiguous()
        coords1 = coords0.clone()
        ub, enc, gru = self.update_block, self.update_block.encoder, self.update_block.gru
        corr = torch.zeros((M, self.CORR_PAD), dtype=torch.float32, device=dev)
        flow = torch.empty((M, 4), dtype=torch.float32, device=dev)
        g = dict(n_img=P, hi=h, wi=w)
        for _ in range(iters):
            ops.corr_lookup_f32(levels, strides, hs, ws, coords1, corr)
            ops.axpby_f32(coords1, coords0, 1.0, -1.0, out=flow)
            cor = ops.conv_gemm_f32(corr, ub.pk("convc1", enc.convc1, cin_pad_to=self.CORR_PAD), act=1, **g)
            cor = ops.conv_gemm_f32(cor, ub.pk("convc2", enc.convc2), act=1, **g)
            flo = ops.conv_gemm_f32(flow, ub.pk("convf1", enc.convf1), act=1, **g)
            flo = ops.conv_gemm_f32(flo, ub.pk("convf2", enc.convf2), act=1, **g)
            ops.conv_gemm_f32(cor, ub.pk("conv", enc.conv), a2=flo, act=1, out=X[:, 128:256], **g)   # 126 ch + 2 zero cols
            ops.copy_cols_f32(flow, 0, X, 254, 2)
            for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
                zr = ops.conv_gemm_f32(net, ub.pk_zr(sfx), a2=X, pad=pad, act=2, **g)
                rh = ops.gru_rh_f32(zr, net)
                q = ops.conv_gemm_f32(rh, ub.pk("q" + sfx, getattr(gru, "convq" + sfx)), a2=X, pad=pad, act=4, **g)
                ops.gru_blend_f32(zr, q, net)
            d = ops.conv_gemm_f32(net, ub.pk("fh1", ub.flow_head.conv1), act=1, **g)
            d = ops.conv_gemm_f32(d, ub.pk("fh2", ub.flow_head.conv2), **g)             # [M][4], cols 2..3 zero
            ops.axpby_f32(coords1, d, 1.0, 1.0, out=coords1)
        m1 = ops.conv_gemm_f32(net, ub.pk("m0", ub.mask[0]), act=1, **g)
        mask = ops.conv_gemm_f32(m1, ub.pk("m2", ub.mask[2]), out_scale=0.25, **g)
        ops.axpby_f32(coords1, coords0, 1.0, -1.0, out=flow)
        up = ops.convex_upsample_f32(flow, mask, P, h, w)                                # (P,2,H,W)
        return up[:t - 1].contiguous(), up[t - 1:].contiguous()

    @E.guarded
    def forward(self, image1, image2, iters=12, flow_init=None, test_mode=True):
        """Reference signature: batched pairs (N,3,H,W) -> (low-res flow is not exposed here, flow_up)."""
        if flow_init is not None or not test_mode:
            raise NotImplementedError("only test_mode=True without flow_init is on the hot path")
        outs = []
        for a, b in zip(image1, image2):
            fwd, _ = self.flows_bidirectional(torch.stack([a, b]), iters=iters)
            outs.append(fwd[0])
        return None, torch.stack(outs)
