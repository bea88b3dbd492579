"""Offline (CPU) numerics of the UNet residual-stream precision: the product UNetVideoModel on the torch stand-ins of
tests/cpu_ops.py (kernel contracts: fp32 arithmetic, every STORED fp16 tensor rounded) against the reference's fp32 output
of the full-width fixture unet_full_t8_64.  Decides whether an fp16-operand / fp32-stream UNet reaches the stated 1e-3
before any kernel work (VERDICT r2 next #1c).

    python tools/stream_numerics.py [f16] [f32] [f32-branch16] [f32+gn] [f32+ln] [f32+proj] [f32+geglu] [f32+attn] [f32+tail]

`f32+<class>`: sensitivity study — the fp32-stream mode with ONE class of fp16 MFMA operands left unrounded (not something
a kernel can do; it shows where the remaining error comes from): gn / ln = GroupNorm / LayerNorm outputs, proj = q|k|v and
text k|v projections, geglu = the feed-forward's gated activation, attn = attention outputs, tail = block outputs that are
only read as operands (last ff-down of a transformer, TemporalModule3D's tail block).
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "tests", "upscale-a-video_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import cpu_ops  # noqa: E402
import golden_cases as GC  # noqa: E402
import synth  # noqa: E402


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


class _FakeHalf(torch.Tensor):
    """fp32 tensor that reports dtype float16: lets an UNROUNDED operand through the stand-ins' dtype asserts."""
    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t.float())

    @property
    def dtype(self):
        return torch.float16


def main():
    from models_video.unet_video import UNetVideoModel
    from uav import engine as E
    torch.set_num_threads(int(os.environ.get("UAV_THREADS", "4")))
    from uav import configs
    cfg = dict(configs.UNET_VIDEO)
    unet = UNetVideoModel.from_config(dict(cfg)).eval()
    unet.load_state_dict(synth.synth_state_dict(unet.state_dict(), seed=1234), strict=True)
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "unet_full_t8_64.pt"))
    sample, low, ehs, ts, cl = GC.unet_inputs(*GC.FULL_CASES["unet_full_t8_64"], cfg["cross_attention_dim"])
    modes = sys.argv[1:] or ["f16", "f32", "f32-branch16"]
    cpu_ops.install()
    from uav import ops
    base = {k: getattr(ops, k) for k in ("groupnorm", "layernorm", "conv_gemm", "attention", "temporal_attention", "cast_f16")}
    real_h = cpu_ops._h
    real_assert = None

    def exact(fn):
        def wrap(*a, **kw):
            cpu_ops._h = lambda x: x.float()
            try:
                return fn(*a, **kw)
            finally:
                cpu_ops._h = real_h
        return wrap

    def conv_exact(kinds):
        def wrap(a1, wt, **kw):
            fp16_out = not kw.get("out_f32", False)
            kind = None
            if fp16_out:
                kind = "geglu" if wt.geglu else ("tail" if kw.get("residual") is not None else "proj")
            a1 = a1 if a1.dtype == torch.float16 else _FakeHalf(a1)
            if kw.get("a2") is not None and kw["a2"].dtype != torch.float16:
                kw["a2"] = _FakeHalf(kw["a2"])
            if kind in kinds and kw.get("out") is None:
                r = exact(base["conv_gemm"])(a1, wt, **kw)
            else:
                r = base["conv_gemm"](a1, wt, **kw)
            return r.as_subclass(torch.Tensor) if isinstance(r, _FakeHalf) else r
        return wrap
    try:
        for mode in modes:
            unet.stream_dtype = torch.float32 if mode.startswith("f32") else torch.float16      # "f32-bf16": fp32 stream, bf16 operands
            E.BRANCH_F32 = "branch16" not in mode
            E.TOKEN_F32 = "-tok16" not in mode
            E.TAIL_HILO = "-hilotail" in mode
            # "-only:up8" / "-only:down64" ...: hi|lo on ONE sampler (kind + input height at this 64x64 clip)
            E.SAMPLER_HILO_SKIP = set()
            if "-only:" in mode:
                keep = mode.split("-only:")[1].split("-")[0]
                allk = [("down", 64), ("down", 32), ("down", 16), ("up", 8), ("up", 16), ("up", 32)]
                E.SAMPLER_HILO_SKIP = {k for k in allk if f"{k[0]}{k[1]}" != keep}
            E.SAMPLER_HILO = "down" if "-hilodown" in mode else "up" if "-hiloup" in mode else ("-hilosamp" in mode)   # product knob
            E.TOKEN_F32_MAX_HW = 256 if "toponly16" in mode else 0      # 64x64 input: levels 32x32 / 16x16 / 8x8 tokens per frame
            keep = set(mode.split("+")[1:])
            cpu_ops._h = real_h
            if "bf16" in mode:                   # bf16 MFMA operands (8-bit mantissa): every stored operand AND the weights
                import copy
                cpu_ops._h = lambda x: x.to(torch.bfloat16).to(torch.float16)
                wcache = {}

                def conv_bf16(a1, wt, **kw):
                    w2 = wcache.get(id(wt))
                    if w2 is None:
                        w2 = copy.copy(wt); w2.w = wt.w.to(torch.bfloat16).to(torch.float16); wcache[id(wt)] = w2
                    return base["conv_gemm"](a1, w2, **kw)
                ops.conv_gemm = cpu_ops.conv_gemm = conv_bf16
            for k, f in base.items():
                setattr(ops, k, f)
            cpu_ops.conv_gemm = base["conv_gemm"]
            if keep:
                if "gn" in keep: ops.groupnorm = exact(base["groupnorm"])
                if "ln" in keep: ops.layernorm = exact(base["layernorm"])
                if "attn" in keep:
                    ops.attention = exact(base["attention"]); ops.temporal_attention = exact(base["temporal_attention"])
                ops.conv_gemm = cpu_ops.conv_gemm = conv_exact(keep)
                if "cast" in keep or "samp" in keep:   # stream -> operand roundings of the down / up sampler inputs
                    ops.cast_f16 = lambda x: x if x.dtype == torch.float16 else _FakeHalf(x)
                if "cast" in keep or "rawx" in keep:   # ... and of the shortcut-conv inputs (the raw copy of the norm1 pass)
                    gn_fn = ops.groupnorm

                    def gn_raw_exact(x1, gamma, beta, **kw):
                        if not kw.get("want_raw"):
                            return gn_fn(x1, gamma, beta, **kw)
                        y, _ = gn_fn(x1, gamma, beta, **kw)
                        x2 = kw.get("x2")
                        if x2 is not None and x2.shape[0] * 2 == x1.shape[0]:
                            x2 = torch.cat([x2, x2])
                        return y, _FakeHalf(x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=-1))
                    ops.groupnorm = gn_raw_exact
            # "-gnin16" / "-lnin16": GroupNorm / LayerNorm passes read the fp32 stream through an fp16 rounding (what a stream
            # stored as an fp16 hi|lo plane pair would allow: the norm passes read the hi plane only, 2 instead of 4 B/element)
            def rounded_input(fn):
                def wrap(x, *a, **kw):
                    xr = x.half().float() if x.dtype == torch.float32 else x
                    kwr = dict(kw)
                    if kw.get("x2") is not None and kw["x2"].dtype == torch.float32:
                        kwr["x2"] = kw["x2"].half().float()
                    if kw.get("want_raw"):               # the raw hi|lo operand copy still comes from the exact stream
                        return fn(xr, *a, **kwr)[0], fn(x, *a, **kw)[1]
                    return fn(xr, *a, **kwr)
                return wrap
            if "-gnin16" in mode:
                ops.groupnorm = rounded_input(ops.groupnorm)
            if "-lnin16" in mode:
                ops.layernorm = rounded_input(ops.layernorm)
            # "-lnfold1" / "-lnfold2": LayerNorm folded into the consuming projection.  1 = the round-3 form (operand fp16(x), the
            # row mean costs mantissa bits); 2 = the operand is fp16(x - m_c), m_c the mean of the row's 128-column chunk (what the
            # producing wave tile knows), the consumer's epilogue adds (m_c - mu) . colsum_c back: emulated here as the UNROUNDED
            # operand rstd * (fp16(x - m_c) + (m_c - mu)) * gamma + beta (the fp16(W * gamma) weight rounding is not emulated)
            if "-lnfold" in mode:
                v2 = "-lnfold2" in mode

                def ln_folded(x, gamma, beta, eps=1e-5):
                    if x.dtype != torch.float32:
                        return base["layernorm"](x, gamma, beta, eps)
                    xf = x.float()
                    mu = xf.mean(-1, keepdim=True)
                    rstd = (xf.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
                    if v2:
                        c = xf.shape[-1]
                        xc = xf.reshape(xf.shape[:-1] + (c // 128, 128))
                        mc = xc.mean(-1, keepdim=True)
                        xr = ((xc - mc).half().float() + mc).reshape(xf.shape)
                    else:
                        xr = xf.half().float()
                    return _FakeHalf((xr - mu) * rstd * gamma.float() + beta.float())
                ops.layernorm = ln_folded
                ops.conv_gemm = cpu_ops.conv_gemm = conv_exact(set())      # strips the _FakeHalf subclass from the results
            t0 = time.time()
            with torch.no_grad():
                out = unet(sample.half(), ts, low.half(), encoder_hidden_states=ehs.half(), class_labels=cl).sample
            print(f"{mode:14s} vs reference fp32 {rel_l2(out, gold['fp32']):.3e}   vs reference fp16 {rel_l2(out, gold['fp16']):.3e}"
                  f"   (reference fp16 vs fp32 {rel_l2(gold['fp16'], gold['fp32']):.3e})   {time.time() - t0:.0f} s", flush=True)
    finally:
        cpu_ops._h = real_h
        cpu_ops.conv_gemm = base["conv_gemm"]
        cpu_ops.restore()


if __name__ == "__main__":
    main()
