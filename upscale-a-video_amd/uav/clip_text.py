"""MI355X-native CLIP text encoder: the `text_encoder` the pipeline calls in `_encode_prompt`
(reference pipeline_upscale_a_video.py:177-321 -> transformers==4.28.1 `CLIPTextModel`, un-vendored; ViT-H/14 text
tower in the release: 24 layers, width 1024, 16 heads, 77 tokens, run twice per pipeline call).

Same arithmetic as `transformers.models.clip.modeling_clip.CLIPTextModel.forward` (last_hidden_state):
token + position embedding -> N x [LayerNorm -> causal self-attention (q,k,v,out with bias, scale d^-1/2) -> +residual
-> LayerNorm -> fc1 -> GELU (erf) | quick-GELU -> fc2 -> +residual] -> final LayerNorm, on the kernels the UNet already
uses: `uav_layernorm_f16`, `uav_conv_gemm_f16` (fused q|k|v projection, bias / activation / residual epilogues) and
`uav_attention_f16` with its causal flag.  Parameters keep the Hugging Face names, so a CLIPTextModel state dict (with or
without the 4.x `text_model.` prefix) loads with strict=True; `from_hf(model)` converts a loaded model.
"""
import types

import torch
import torch.nn as nn

from . import engine as E
from . import ops


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.k_proj = nn.Linear(c, c); self.v_proj = nn.Linear(c, c); self.q_proj = nn.Linear(c, c); self.out_proj = nn.Linear(c, c)


class _MLP(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.fc1 = nn.Linear(c, inner); self.fc2 = nn.Linear(inner, c)


class _Layer(nn.Module):
    def __init__(self, c, inner, eps):
        super().__init__()
        self.self_attn = _Attn(c)
        self.layer_norm1 = nn.LayerNorm(c, eps=eps)
        self.mlp = _MLP(c, inner)
        self.layer_norm2 = nn.LayerNorm(c, eps=eps)


class _Embeddings(nn.Module):
    def __init__(self, vocab, c, positions):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, c)
        self.position_embedding = nn.Embedding(positions, c)


class _Encoder(nn.Module):
    def __init__(self, n, c, inner, eps):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c, inner, eps) for _ in range(n)])


class UavCLIPTextModel(E.EngineModule):
    def __init__(self, vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, config=None):
        super().__init__()
        if hidden_act not in ("gelu", "quick_gelu"):
            raise NotImplementedError(f"CLIP hidden_act {hidden_act!r}")
        if hidden_size % num_attention_heads or hidden_size // num_attention_heads not in (64, 128):
            raise NotImplementedError("head_dim must be 64 or 128")
        self.heads, self.act = num_attention_heads, hidden_act
        self.embeddings = _Embeddings(vocab_size, hidden_size, max_position_embeddings)
        self.encoder = _Encoder(num_hidden_layers, hidden_size, intermediate_size, layer_norm_eps)
        self.final_layer_norm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.config = config if config is not None else types.SimpleNamespace(
            vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
            num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
            max_position_embeddings=max_position_embeddings, hidden_act=hidden_act, layer_norm_eps=layer_norm_eps)

    @classmethod
    def from_hf(cls, hf):
        """Build from a loaded `transformers.CLIPTextModel` (its parameters are copied; dtype and device kept)."""
        cfg = hf.config
        m = cls(cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                cfg.max_position_embeddings, cfg.hidden_act, cfg.layer_norm_eps, config=cfg)
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in hf.state_dict().items()
              if not k.endswith("position_ids")}
        p0 = next(hf.parameters())
        m = m.to(device=p0.device, dtype=p0.dtype)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()
              if not k.endswith("position_ids")}
        return super().load_state_dict(sd, strict=strict, **kw)

    @property
    def dtype(self):
        return self.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.final_layer_norm.weight.device

    @E.guarded
    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, **kw):
        if attention_mask is not None:
            raise NotImplementedError("the pipeline passes attention_mask=None (config.use_attention_mask is unset for CLIP)")
        dev = E._dev(self.final_layer_norm.weight)
        ids = input_ids.to(dev).reshape(-1, input_ids.shape[-1])
        b, l = ids.shape
        c = self.final_layer_norm.weight.numel()
        d = c // self.heads
        tok = E.f16_param(self, "tok", self.embeddings.token_embedding.weight)
        pos = E.f16_param(self, "pos", self.embeddings.position_embedding.weight)
        # embedding lookup = row gather; the sum is rounded to fp16 once (the fp16 reference model does the same)
        x = (tok.index_select(0, ids.reshape(-1)).float() + pos[:l].float().repeat(b, 1)).half().contiguous()
        for i, layer in enumerate(self.encoder.layers):
            a = layer.self_attn
            h = E.layer_norm(self, f"l{i}.ln1", layer.layer_norm1, x)
            qkv = ops.linear(h, E.packed_cat(self, f"l{i}.qkv", [a.q_proj, a.k_proj, a.v_proj]))
            o = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], bq=b, lq=l, lk=l, heads=self.heads, head_dim=d,
                              scale=d ** -0.5, q_stride=3 * c, k_stride=3 * c, v_stride=3 * c, causal=True)
            x = ops.linear(o, E.packed_conv(self, f"l{i}.out", a.out_proj), residual=x)
            h = E.layer_norm(self, f"l{i}.ln2", layer.layer_norm2, x)
            h = ops.linear(h, E.packed_conv(self, f"l{i}.fc1", layer.mlp.fc1), act=self.act)
            x = ops.linear(h, E.packed_conv(self, f"l{i}.fc2", layer.mlp.fc2), residual=x)
        x = E.layer_norm(self, "final", self.final_layer_norm, x)
        out = x.reshape(b, l, c).to(self.dtype)
        return _Output(last_hidden_state=out)


class _Output(tuple):
    """`outputs[0]` / `.last_hidden_state`, like transformers' BaseModelOutputWithPooling for what the pipeline reads."""

    def __new__(cls, last_hidden_state):
        t = super().__new__(cls, (last_hidden_state,))
        t.last_hidden_state = last_hidden_state
        return t
