"""MI355X-native transformer blocks of Upscale-A-Video (drop-in for the reference's
`models_video/attention.py`: CrossAttention :44, Transformer3DModel :292, BasicTransformerBlock
:414, TemporalAttention :626, RelativePositionBias :735).

State-dict keys and constructor arguments follow the reference; execution is channels-last on the
HIP kernel library:
  * q/k/v projections are fused GEMMs (`uav_conv_gemm_f16`), to_out carries bias + residual in
    its epilogue, the GEGLU feed-forward gates in the up-projection's epilogue;
  * spatial self-attention / text cross-attention run in the flash kernel `uav_attention_f16`
    (scores never reach HBM; reference attention.py:214-234 materialises them);
  * text K/V are projected ONCE per prompt tensor (the reference recomputes them per frame and per
    step, attention.py:364,177-178);
  * temporal attention reads the (B,T,H,W,C) rows in place (`uav_temporal_attention_f16`): the
    reference's (b f) d c <-> (b d) f c transposes (attention.py:555,560) do not exist here,
    LayerNorm and the projections are per-token and need no re-ordering.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn

from uav import engine as E
from uav import ops

from ._compat import BaseOutput, ConfigMixin, ModelMixin, register_to_config
from .resnet import ResnetBlock3DCNN

TEXT_KV_ENTRIES = 4        # prompt tensors whose text K/V a block keeps (positive + negative prompt of two callers)


@dataclass
class Transformer3DModelOutput(BaseOutput):
    sample: torch.FloatTensor


class RotaryEmbedding(nn.Module):
    """Parameter container of rotary-embedding-torch's RotaryEmbedding(dim): `freqs` appears in the
    reference state dict (one shared instance, unet_video.py:203)."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        self.dim = dim
        self.freqs = nn.Parameter(1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)), requires_grad=False)


class RelativePositionBias(nn.Module):
    """T5-style bucketed relative position bias (reference attention.py:735-772)."""

    def __init__(self, heads=8, num_buckets=32, max_distance=128):
        super().__init__()
        self.num_buckets, self.max_distance = num_buckets, max_distance
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)

    def bucket_table(self, n):
        """Integer bucket index [n][n] (host-side index math)."""
        nb = self.num_buckets // 2
        max_exact = nb // 2
        tab = torch.zeros((n, n), dtype=torch.long)
        for i in range(n):
            for j in range(n):
                rel = j - i
                m = -rel
                ret = nb if m < 0 else 0
                m = abs(m)
                if m < max_exact:
                    ret += m
                else:
                    v = max_exact + int(math.log(m / max_exact) / math.log(self.max_distance / max_exact) * (nb - max_exact))
                    ret += min(v, nb - 1)
                tab[i, j] = ret
        return tab


class CrossAttention(E.EngineModule):
    """Spatial self-attention (cross_attention_dim=None) or text cross-attention."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, added_kv_proj_dim=None, norm_num_groups=None,
                 use_relative_position=False):
        super().__init__()
        if added_kv_proj_dim is not None or norm_num_groups is not None or use_relative_position:
            raise NotImplementedError("not used by the released UNet config")
        inner = dim_head * heads
        self.is_cross = cross_attention_dim is not None
        cad = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self._use_memory_efficient_attention_xformers = False
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cad, inner, bias=bias)
        self.to_v = nn.Linear(cad, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def run(self, x, residual, *, bq, lq, text=None, q_per_kv=1, ln=None):
        """x: normalised tokens [bq*lq][C] — or, with `ln` (the block's LayerNorm in front of this attention), the
        un-normalised stream itself: the LayerNorm is then folded into the q / q|k|v projection where the launch allows it
        (engine.ln_linear); text: TextKV cache entry (k, v, lk) for cross-attention."""
        c = self.heads * self.dim_head
        if self.is_cross:
            k, v, lk = text[:3]
            kvp = text[3] if len(text) > 3 else None
            sub = self.fused_params(x, ln, text, rows_per_kv=q_per_kv * lq) if residual is x else None
            if sub is not None:
                # the whole sub-layer in one launch (csrc/xattn_fused.hip): the fp32 stream is read once and written once
                return ops.xattn_sublayers(x, [sub], rows_per_kv=q_per_kv * lq, lk=lk, scale=self.scale)
            q = ops.linear(x, E.packed_conv(self, "q", self.to_q)) if ln is None else E.ln_linear(self, "q", ln, x, [self.to_q])
            o = ops.attention(q, k, v, bq=bq, lq=lq, lk=lk, heads=self.heads, head_dim=self.dim_head, q_per_kv=q_per_kv,
                              scale=self.scale, q_stride=c, k_stride=2 * c, v_stride=2 * c)
        else:
            qkv = ops.linear(x, E.packed_cat(self, "qkv", [self.to_q, self.to_k, self.to_v])) if ln is None else \
                E.ln_linear(self, "qkv", ln, x, [self.to_q, self.to_k, self.to_v])
            o = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], bq=bq, lq=lq, lk=lq, heads=self.heads,
                              head_dim=self.dim_head, scale=self.scale, q_stride=3 * c, k_stride=3 * c, v_stride=3 * c)
        s32 = residual.dtype == torch.float32
        return ops.linear(o, E.packed_conv(self, "out", self.to_out[0]), residual=residual, out_f32=s32, ln_produce=s32 and E.LN_FOLD)

    def fused_params(self, x, ln, text, *, rows_per_kv):
        """(gamma, beta, eps, W_q, K | V, W_out, bias) of this sub-layer for the fused kernel (ops.xattn_sublayers), or None where it does
        not apply: text cross-attention on an fp32 stream of 512 channels / 8 heads, <= 96 keys, whole 128-row tiles per kv batch."""
        if not self.is_cross or ln is None or len(text) < 4 or text[3] is None or not E.XATTN_FUSED or E.LN_FOLD:
            return None
        if self.to_q.bias is not None or self.to_out[0].bias is None:
            return None
        if not ops.xattn_ok(x, heads=self.heads, head_dim=self.dim_head, lk=text[2], rows_per_kv=rows_per_kv):
            return None
        wq = self._cache().get(("xattn", "q"), lambda: ops.pack_xattn_weight(self.to_q.weight, "q", E._dev(self.to_q.weight)), (self.to_q.weight,))
        wo = self._cache().get(("xattn", "out"), lambda: ops.pack_xattn_weight(self.to_out[0].weight, "out", E._dev(self.to_q.weight)),
                               (self.to_out[0].weight,))
        return (E.f32_param(self, "xattn.g", ln.weight), E.f32_param(self, "xattn.b", ln.bias), ln.eps, wq, text[3], wo,
                E.f32_param(self, "xattn.ob", self.to_out[0].bias))

    def project_text(self, ehs_rows, n_text=None):
        """K|V of the text tokens: [B*77][2C] (fused GEMM), computed once per prompt tensor — and, where the fused sub-layer kernel
        takes the block (512 channels, 8 heads), the same values in its MFMA-fragment stream order (third entry, else None)."""
        kv = ops.linear(ehs_rows, E.packed_cat(self, "kv", [self.to_k, self.to_v]))
        c = self.heads * self.dim_head
        k, v = kv[:, :c], kv[:, c:]
        packed = None
        if (E.XATTN_FUSED and n_text and c == ops.XATTN_C and self.heads == ops.XATTN_HEADS and n_text <= ops.XATTN_MAX_KEYS
                and ehs_rows.shape[0] % n_text == 0):
            packed = ops.xattn_pack_kv(k, v, n_batch=ehs_rows.shape[0] // n_text, lk=n_text, k_stride=2 * c, v_stride=2 * c)
        return k, v, packed


class TemporalAttention(CrossAttention):
    """Per-pixel attention over the frame axis with RoPE + relative position bias
    (reference attention.py:626-733)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, added_kv_proj_dim=None, norm_num_groups=None,
                 rotary_emb=None):
        super().__init__(query_dim, cross_attention_dim, heads, dim_head, dropout, bias, upcast_attention,
                         upcast_softmax, added_kv_proj_dim, norm_num_groups)
        self.time_rel_pos_bias = RelativePositionBias(heads=heads, max_distance=32)
        self.rotary_emb = rotary_emb

    def _tables(self, t_len):
        def build():
            dev = E._dev(self.to_q.weight)
            tab = self.time_rel_pos_bias.bucket_table(t_len).to(dev)
            w = self.time_rel_pos_bias.relative_attention_bias.weight.detach().float()
            bias = w[tab].permute(2, 0, 1).contiguous()                          # (heads, T, T)
            if self.rotary_emb is not None:
                fr = self.rotary_emb.freqs.detach().float().to(dev)
                ang = torch.arange(t_len, device=dev, dtype=torch.float32)[:, None] * fr[None, :]
                return bias, ang.cos().contiguous(), ang.sin().contiguous(), 2 * fr.numel()
            return bias, None, None, 0
        return self._cache().get(("ttab", t_len), build, (self.time_rel_pos_bias.relative_attention_bias.weight,
                                                          None if self.rotary_emb is None else self.rotary_emb.freqs))

    def fused_temporal_params(self, x, ln, g: E.Geom):
        """(gamma, beta, eps, W_q, W_k, W_v, W_out packed, bias, rel-pos bias, RoPE cos, sin, rot_dim) for the fused temporal sub-layer kernel
        (ops.tattn_sublayer / ops.block_attn_sublayers), or None where it does not apply (fp32 stream of 512 channels / 8 heads, T = 8,
        rot_dim 32, whole 16-pixel tiles)."""
        bias, cos, sin, rot = self._tables(g.t)
        if (ln is None or not E.TATTN_FUSED or E.LN_FOLD or self.to_q.bias is not None or self.to_out[0].bias is None or cos is None
                or not ops.tattn_ok(x, heads=self.heads, head_dim=self.dim_head, t_len=g.t, hw=g.hw, rot_dim=rot)):
            return None
        dev = E._dev(self.to_q.weight)
        pk = lambda name, lin, kind: self._cache().get(("tattn", name), lambda: ops.pack_xattn_weight(lin.weight, kind, dev), (lin.weight,))
        return (E.f32_param(self, "tattn.g", ln.weight), E.f32_param(self, "tattn.b", ln.bias), ln.eps,
                pk("q", self.to_q, "q"), pk("k", self.to_k, "q"), pk("v", self.to_v, "q"), pk("o", self.to_out[0], "out"),
                E.f32_param(self, "tattn.ob", self.to_out[0].bias), bias, cos, sin, rot)

    def run_temporal(self, x, residual, g: E.Geom, ln=None, next_ln=None):
        c = self.heads * self.dim_head
        bias, cos, sin, rot = self._tables(g.t)
        tp = self.fused_temporal_params(x, ln, g) if residual is x else None
        if tp is not None:
            # the whole sub-layer in one launch (csrc/xattn_fused.hip, tattn_sublayer_kernel): LayerNorm -> q | k | v -> RoPE + bias + per-pixel
            # softmax over the frames -> to_out -> + residual, the fp32 stream read once and written once
            return ops.tattn_sublayer(x, *tp[:11], n_batch=g.b, t_len=g.t, hw=g.hw, rot_dim=tp[11], scale=self.scale, next_ln=next_ln)
        qkv = ops.linear(x, E.packed_cat(self, "qkv", [self.to_q, self.to_k, self.to_v])) if ln is None else \
            E.ln_linear(self, "qkv", ln, x, [self.to_q, self.to_k, self.to_v])
        o = ops.temporal_attention(qkv, n_batch=g.b, t_len=g.t, hw=g.hw, c=c, heads=self.heads, scale=self.scale,
                                   rope_cos=cos, rope_sin=sin, rot_dim=rot, bias=bias)
        s32 = residual.dtype == torch.float32
        return ops.linear(o, E.packed_conv(self, "out", self.to_out[0]), residual=residual, out_f32=s32, ln_produce=s32 and E.LN_FOLD)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(E.EngineModule):
    """diffusers FeedForward(dim, activation_fn='geglu') (spec: diffusers_attention.py:735-823)."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu"):
        super().__init__()
        if activation_fn != "geglu":
            raise NotImplementedError(activation_fn)
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])

    def fused_params(self, x, ln):
        """(gamma, beta, eps, packed W_up | W_down fragments, b_up, b_down) for the fused feed-forward kernel (ops.ff_sublayer), or None where it
        does not apply (fp32 stream of 512 channels, 2 048 hidden channels, whole 128-row tiles)."""
        up, down = self.net[0].proj, self.net[2]
        if (ln is None or not E.FF_FUSED or E.LN_FOLD or up.bias is None or down.bias is None or up.out_features != 2 * down.in_features
                or not ops.ff_ok(x, inner=down.in_features)):
            return None
        dev = E._dev(up.weight)
        w = self._cache().get(("ff", "w"), lambda: ops.pack_ff_weights(up.weight, down.weight, dev), (up.weight, down.weight))
        return (E.f32_param(self, "ff.g", ln.weight), E.f32_param(self, "ff.b", ln.bias), ln.eps, w,
                E.f32_param(self, "ff.ub", up.bias), E.f32_param(self, "ff.db", down.bias))

    def run(self, x, residual, out_f32=None, ln=None, out_hilo=False):
        """out_hilo (fp32 stream): the fp32 sum leaves as the fp16 hi | lo operand pair of proj_out ([M][2 C], engine.tail_hilo)."""
        if residual is x and out_f32 is not False:
            fp = self.fused_params(x, ln)
            if fp is not None:      # the whole sub-layer in one launch (csrc/xattn_fused.hip, ff_sublayer_kernel)
                return ops.ff_sublayer(x, *fp, out_f32=not out_hilo, out_hilo=out_hilo)
        h = ops.linear(x, E.packed_conv(self, "up", self.net[0].proj, geglu=True)) if ln is None else \
            E.ln_linear(self, "up", ln, x, [self.net[0].proj], geglu=True)
        if out_hilo:
            return ops.linear(h, E.packed_conv(self, "down", self.net[2]), residual=residual, out_f32=True, out_hilo=True)
        return ops.linear(h, E.packed_conv(self, "down", self.net[2]), residual=residual,
                          out_f32=(residual.dtype == torch.float32) if out_f32 is None else out_f32)


class BasicTransformerBlock(E.EngineModule):
    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, attention_bias=False, only_cross_attention=False,
                 upcast_attention=False, use_first_frame=False, use_relative_position=False, rotary_emb=None):
        super().__init__()
        if num_embeds_ada_norm is not None or use_first_frame:
            raise NotImplementedError("AdaLayerNorm / SparseCausalAttention are not used by the released config")
        self.only_cross_attention = only_cross_attention
        self.attn1 = CrossAttention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                                    bias=attention_bias, cross_attention_dim=cross_attention_dim if only_cross_attention else None)
        self.norm1 = nn.LayerNorm(dim)
        if cross_attention_dim is not None:
            self.attn2 = CrossAttention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                                        dim_head=attention_head_dim, dropout=dropout, bias=attention_bias)
            self.norm2 = nn.LayerNorm(dim)
        else:
            self.attn2, self.norm2 = None, None
        self.attn_temporal = TemporalAttention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                                               dropout=dropout, bias=attention_bias, rotary_emb=rotary_emb)
        nn.init.zeros_(self.attn_temporal.to_out[0].weight.data)
        self.norm_temporal = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self.norm3 = nn.LayerNorm(dim)

    def _text_kv(self, attn, ehs_rows, tag, n_text=None):
        """K/V of the text tokens, cached per prompt tensor: the cache holds a reference to
        `ehs_rows` (so its storage cannot be recycled) and is keyed on identity + version."""
        c = self._cache()
        hits = c.store.get(("textkv", tag), ())
        wstamp = E._stamp((attn.to_k.weight, attn.to_v.weight))      # in-place weight edits invalidate the projection too
        for hit in hits:
            if hit[0] is ehs_rows and hit[1] == ehs_rows._version and hit[3] == wstamp:
                return E.acquire(hit[2])
        kv = E.publish(attn.project_text(ehs_rows, n_text))
        # a few entries: the guidance branches evaluated one by one (pipeline.shard_cfg / overlap_streams) alternate between
        # two prompt tensors
        c.store[("textkv", tag)] = ((ehs_rows, ehs_rows._version, kv, wstamp),) + tuple(hits)[:TEXT_KV_ENTRIES - 1]
        return kv

    def whole_block_params(self, x, g: E.Geom, ehs_rows, n_text):
        """(cross pair, temporal params, feed-forward params) when the whole block runs as ONE launch on fp32 rows shaped like x
        (ops.block_sublayers), else None."""
        if not (self.only_cross_attention and self.attn2 is not None and E.XATTN_PAIR and E.BLOCK_ATTN_FUSED and E.BLOCK_FF_FUSED):
            return None
        ffp = self.ff.fused_params(x, self.norm3)
        if ffp is None or self.attn1.scale != self.attn2.scale:
            return None
        k, v, kvp = self._text_kv(self.attn1, ehs_rows, "a1", n_text)
        s1 = self.attn1.fused_params(x, self.norm1, (k, v, n_text, kvp), rows_per_kv=g.t * g.hw)
        if s1 is None:
            return None
        k, v, kvp = self._text_kv(self.attn2, ehs_rows, "a2", n_text)
        s2 = self.attn2.fused_params(x, self.norm2, (k, v, n_text, kvp), rows_per_kv=g.t * g.hw)
        tp = self.attn_temporal.fused_temporal_params(x, self.norm_temporal, g) if s2 is not None else None
        return None if tp is None else ([s1, s2], tp, ffp)

    def run(self, x, g: E.Geom, ehs_rows, n_text, out_f32=None, out_hilo=False):
        """x: tokens [B*T*HW][C] (rows ordered b,t,p), fp16 or fp32 (residual stream); ehs_rows: [B*n_text][Cx] fp16.
        out_f32=False: the block's output is only read as an MFMA operand (proj_out) -> written in fp16;
        out_hilo: ... as the fp16 hi | lo pair of the fp32 rows instead (FeedForward.run)."""
        bq, lq = g.n_img, g.hw
        # every LayerNorm is handed to its consumer together with the un-normalised stream: engine.ln_linear folds it into
        # the projection when the stream is fp32 and its producer wrote the operand copy, else runs the LayerNorm pass
        t1 = t2 = None
        if self.only_cross_attention:
            k, v, kvp = self._text_kv(self.attn1, ehs_rows, "a1", n_text)
            t1 = (k, v, n_text, kvp)
        if self.attn2 is not None:
            k, v, kvp = self._text_kv(self.attn2, ehs_rows, "a2", n_text)
            t2 = (k, v, n_text, kvp)
        # (the feed-forward as one launch does its own LayerNorm: norm3 is then not asked of the attention kernels' epilogue)
        ff_fused = out_f32 is not False and self.ff.fused_params(x, self.norm3) is not None
        pair = None
        if t1 is not None and t2 is not None and E.XATTN_PAIR:
            # attn1 (only_cross_attention) and attn2 are both text cross-attention: ONE launch for the two sub-layers, the rows between
            # them never leave the accumulators (csrc/xattn_fused.hip, n_subs = 2)
            s1 = self.attn1.fused_params(x, self.norm1, t1, rows_per_kv=g.t * lq)
            s2 = self.attn2.fused_params(x, self.norm2, t2, rows_per_kv=g.t * lq) if s1 is not None else None
            if s2 is not None and self.attn1.scale == self.attn2.scale:
                pair = (s1, s2)
        if pair is not None and E.BLOCK_ATTN_FUSED:
            # ... and the temporal sub-layer behind them in the same launch (tattn_sublayer_kernel<2>): x is read once and written once for
            # the three attention sub-layers of the block
            tp = self.attn_temporal.fused_temporal_params(x, self.norm_temporal, g)
            if tp is not None and ff_fused and E.BLOCK_FF_FUSED:
                # ... and the feed-forward behind them: the whole block in one launch (tattn_sublayer_kernel<2, 1>)
                return ops.block_sublayers(x, list(pair), tp, self.ff.fused_params(x, self.norm3), n_batch=g.b, t_len=g.t, hw=g.hw, lk=n_text,
                                           cross_scale=self.attn1.scale, temporal_scale=self.attn_temporal.scale,
                                           out_f32=not out_hilo, out_hilo=out_hilo)
            if tp is not None:
                # (+ norm3 of the finished rows for the feed-forward: the same parameter tensors FeedForward.run -> engine.ln_linear looks up)
                nxt = (E.f32_param(self.ff, "up.ln.g", self.norm3.weight), E.f32_param(self.ff, "up.ln.b", self.norm3.bias), self.norm3.eps) \
                    if E.NEXT_LN and not ff_fused else None
                x = ops.block_attn_sublayers(x, list(pair), tp, n_batch=g.b, t_len=g.t, hw=g.hw, lk=n_text, cross_scale=self.attn1.scale,
                                             temporal_scale=self.attn_temporal.scale, next_ln=nxt)
                return self.ff.run(x, x, out_f32, ln=self.norm3, out_hilo=out_hilo)
        if pair is not None:
            x = ops.xattn_sublayers(x, list(pair), rows_per_kv=g.t * lq, lk=n_text, scale=self.attn1.scale)
        else:
            if t1 is not None:
                x = self.attn1.run(x, x, bq=bq, lq=lq, text=t1, q_per_kv=g.t, ln=self.norm1)
            else:
                x = self.attn1.run(x, x, bq=bq, lq=lq, ln=self.norm1)
            if t2 is not None:
                x = self.attn2.run(x, x, bq=bq, lq=lq, text=t2, q_per_kv=g.t, ln=self.norm2)
        nxt = (E.f32_param(self.ff, "up.ln.g", self.norm3.weight), E.f32_param(self.ff, "up.ln.b", self.norm3.bias), self.norm3.eps) \
            if E.NEXT_LN and not ff_fused else None
        x = self.attn_temporal.run_temporal(x, x, g, ln=self.norm_temporal, next_ln=nxt)
        return self.ff.run(x, x, out_f32, ln=self.norm3, out_hilo=out_hilo)


class Transformer3DModel(ModelMixin, ConfigMixin, E.EngineModule):
    @register_to_config
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1, dropout=0.0,
                 norm_num_groups=32, cross_attention_dim=None, attention_bias=False, activation_fn="geglu",
                 num_embeds_ada_norm=None, use_linear_projection=False, only_cross_attention=False,
                 upcast_attention=False, use_first_frame=False, use_relative_position=False, rotary_emb=None):
        super().__init__()
        if not use_linear_projection:
            raise NotImplementedError("the released config uses use_linear_projection=True")
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.resblock_temporal = ResnetBlock3DCNN(in_channels=in_channels, kernel=(3, 1, 1), temb_channels=None)
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                  attention_bias=attention_bias, only_cross_attention=only_cross_attention,
                                  upcast_attention=upcast_attention, rotary_emb=rotary_emb)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(in_channels, inner)

    def run(self, x, g: E.Geom, ehs_rows, n_text):
        x = self.resblock_temporal.run(x, g, None)
        res = x
        s32 = res.dtype == torch.float32         # fp32 residual stream: the token stream is one too (LayerNorm inputs)
        tok32 = s32 and E.TOKEN_F32 and (E.TOKEN_F32_MAX_HW <= 0 or g.hw <= E.TOKEN_F32_MAX_HW)
        if tok32 and E.PROJ_IN_FUSED and len(self.transformer_blocks) == 1 and not E.LN_FOLD and self.proj_in.bias is not None \
                and tuple(self.proj_in.weight.shape) == (x.shape[1], x.shape[1]):
            # GroupNorm apply -> proj_in -> the whole block in ONE launch (tattn_sublayer_kernel<2, 1, 1>): the kernel reads the GroupNorm's
            # input once, normalises with the per-frame scale | shift rows of the statistics finalize and multiplies by proj_in on the way in
            blk = self.transformer_blocks[0]
            wb = blk.whole_block_params(x, g, ehs_rows, n_text)
            if wb is not None:
                sc, sh = ops.groupnorm_scale_shift(x, E.f32_param(self, "norm.g", self.norm.weight), E.f32_param(self, "norm.b", self.norm.bias),
                                                   n_inst=g.n_img, rows_per_inst=g.hw, groups=self.norm.num_groups, eps=self.norm.eps)
                dev = E._dev(self.proj_in.weight)
                wp = self._cache().get(("projin", "w"), lambda: ops.pack_xattn_weight(self.proj_in.weight, "out", dev), (self.proj_in.weight,))
                th = E.tail_hilo()
                tok = ops.block_sublayers(x, wb[0], wb[1], wb[2], n_batch=g.b, t_len=g.t, hw=g.hw, lk=n_text, cross_scale=blk.attn1.scale,
                                          temporal_scale=blk.attn_temporal.scale, out_f32=not th, out_hilo=th,
                                          proj_in=(sc, sh, wp, E.f32_param(self, "proj_in.b", self.proj_in.bias)))
                cw = E.packed_conv_hilo(self, "proj_out", self.proj_out) if th else E.packed_conv(self, "proj_out", self.proj_out)
                if not th:
                    tok = ops.cast_f16(tok)          # (TAIL_HILO off: proj_out reads fp16 rows)
                return ops.linear(tok, cw, residual=res, out_f32=s32, gn_groups=self.norm.num_groups)
        n = E.group_norm(self, "norm", self.norm, x, n_inst=g.n_img, rows_per_inst=g.hw, silu=False)   # per frame
        last = len(self.transformer_blocks) - 1
        tail_hilo = tok32 and E.tail_hilo()
        tok = ops.linear(n, E.packed_conv(self, "proj_in", self.proj_in), out_f32=tok32, ln_produce=tok32 and E.LN_FOLD)
        for i, blk in enumerate(self.transformer_blocks):
            # proj_out reads the last block's output as an operand: fp16, or (TAIL_HILO) the fp32 rows as a hi | lo pair
            tok = blk.run(tok, g, ehs_rows, n_text, out_f32=False if (i == last and not tail_hilo) else None,
                          out_hilo=(i == last and tail_hilo))
        if tail_hilo:          # the last feed-forward wrote the hi | lo pair itself (ops.conv_gemm out_hilo; a cast pass where it cannot)
            return ops.linear(tok, E.packed_conv_hilo(self, "proj_out", self.proj_out), residual=res, out_f32=s32,
                              gn_groups=self.norm.num_groups)
        return ops.linear(tok, E.packed_conv(self, "proj_out", self.proj_out), residual=res, out_f32=s32,
                          gn_groups=self.norm.num_groups)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, return_dict=True):
        rows, g = E.to_rows(hidden_states, c_pad=self.in_channels)
        ehs = encoder_hidden_states.half().reshape(-1, encoder_hidden_states.shape[-1]).contiguous()
        y = self.run(rows, g, ehs, encoder_hidden_states.shape[1])
        out = E.from_rows(y, g, self.in_channels, out_dtype=hidden_states.dtype)
        return Transformer3DModelOutput(sample=out) if return_dict else (out,)
