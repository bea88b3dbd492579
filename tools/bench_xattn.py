"""Same-process, interleaved A/B of the fused text cross-attention sub-layer kernel (csrc/xattn_fused.hip) against the four launches it
replaces (LayerNorm -> to_q -> 77-key attention -> to_out + residual) on the two 512-channel levels of BASELINE configs[1].
One JSON line per shape: median / min ms of the fused launch, of the chain and of each of its four launches; algorithmic TFLOP/s and the
HBM rate of the 8 B per element the fused kernel must move; max |difference| of the two results.
usage: python tools/bench_xattn.py"""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")
ROUNDS, PER = int(os.environ.get("UAV_XA_ROUNDS", "5")), int(os.environ.get("UAV_XA_PER", "6"))
C, H, D, LK = 512, 8, 64, 77


def time_once(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, nb, rpk):
    m = nb * rpk
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(m, C, generator=g) * 1.3 + 0.2).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    wq = (torch.randn(C, C, generator=g) * C ** -0.5).half().float().to(dev); wo = (torch.randn(C, C, generator=g) * C ** -0.5).half().float().to(dev)
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    kv = torch.randn(nb * LK, 2 * C, generator=g).half().to(dev)
    k, v = kv[:, :C], kv[:, C:]
    cq, co = ops.pack_conv(wq, None, device=dev), ops.pack_conv(wo, bo, device=dev)
    wqp, wop = ops.pack_xattn_weight(wq, "q", dev), ops.pack_xattn_weight(wo, "out", dev)
    kvp = ops.xattn_pack_kv(k, v, n_batch=nb, lk=LK, k_stride=2 * C, v_stride=2 * C)
    scale = D ** -0.5
    st = {}

    def ln(): st["n"] = ops.layernorm(x, gamma, beta, 1e-5)
    def toq(): st["q"] = ops.linear(st["n"], cq)
    def att(): st["o"] = ops.attention(st["q"], k, v, bq=nb, lq=rpk, lk=LK, heads=H, head_dim=D, scale=scale, q_stride=C, k_stride=2 * C, v_stride=2 * C)
    def out(): st["y"] = ops.linear(st["o"], co, residual=x, out_f32=True)
    def chain(): ln(); toq(); att(); out()
    def fused(): st["f"] = ops.xattn_sublayer(x, gamma, beta, 1e-5, wqp, kvp, wop, bo, rows_per_kv=rpk, lk=LK, scale=scale)
    sub = (gamma, beta, 1e-5, wqp, kvp, wop, bo)
    def pair(): st["p"] = ops.xattn_sublayers(x, [sub, sub], rows_per_kv=rpk, lk=LK, scale=scale)       # two sub-layers (same weights here) in one launch

    fns = {"fused": fused, "pair": pair, "chain": chain, "layernorm": ln, "to_q": toq, "attention": att, "to_out": out}
    chain(); fused(); pair(); chain(); fused()
    torch.cuda.synchronize()
    t = {kk: [] for kk in fns}
    for _ in range(ROUNDS):
        for kk, f in fns.items():
            t[kk].append(time_once(f, PER))
    fl = 2.0 * m * 2 * C * C + 4.0 * m * LK * C
    d = {"case": name, "rows": m, "max_abs_diff_fused_vs_chain": float((st["f"] - st["y"]).abs().max())}
    for kk in fns:
        d[kk + "_ms"] = {"median": round(statistics.median(t[kk]), 4), "min": round(min(t[kk]), 4)}
    med = d["fused_ms"]["median"]
    d["fused_tflops"] = round(fl / med / 1e9, 1)
    d["fused_algorithmic_GBps"] = round(8.0 * m * C / med / 1e6, 0)
    d["speedup_vs_chain"] = round(d["chain_ms"]["median"] / med, 3)
    d["pair_speedup_vs_two_chains"] = round(2 * d["chain_ms"]["median"] / d["pair_ms"]["median"], 3)
    print(json.dumps(d), flush=True)


def tcase(name, nb, hh, ww):
    """The temporal sub-layer: fused launch against LayerNorm -> q|k|v -> temporal attention -> to_out + residual."""
    T = 8
    hw = hh * ww
    m = nb * T * hw
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(m, C, generator=g) * 1.3 + 0.2).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    ws = [(torch.randn(C, C, generator=g) * C ** -0.5).half().float().to(dev) for _ in range(4)]
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    relb = (torch.randn(H, T, T, generator=g) * 0.5).to(dev).contiguous()
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(T).float()[:, None] * fr[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    cqkv, co = ops.pack_conv(torch.cat(ws[:3], 0), None, device=dev), ops.pack_conv(ws[3], bo, device=dev)
    pk = [ops.pack_xattn_weight(w_, "q", dev) for w_ in ws[:3]] + [ops.pack_xattn_weight(ws[3], "out", dev)]
    scale = D ** -0.5
    st = {}

    def ln(): st["n"] = ops.layernorm(x, gamma, beta, 1e-5)
    def qkv(): st["qkv"] = ops.linear(st["n"], cqkv)
    def att(): st["o"] = ops.temporal_attention(st["qkv"], n_batch=nb, t_len=T, hw=hw, c=C, heads=H, scale=scale, rope_cos=cos, rope_sin=sin, rot_dim=32, bias=relb)
    def out(): st["y"] = ops.linear(st["o"], co, residual=x, out_f32=True)
    def chain(): ln(); qkv(); att(); out()
    def fused(): st["f"] = ops.tattn_sublayer(x, gamma, beta, 1e-5, *pk, bo, relb, cos, sin, n_batch=nb, t_len=T, hw=hw, rot_dim=32, scale=scale)

    # the block launch: two cross-attention sub-layers (synthetic weights) + this temporal sub-layer in one kernel, against pair + temporal
    LK = 77
    kvx = torch.randn(nb * LK, 2 * C, generator=g).half().to(dev)
    kvp = ops.xattn_pack_kv(kvx[:, :C], kvx[:, C:], n_batch=nb, lk=LK, k_stride=2 * C, v_stride=2 * C)
    xs = (gamma, beta, 1e-5, pk[0], kvp, pk[3], bo)
    tp = (gamma, beta, 1e-5, *pk, bo, relb, cos, sin, 32)
    def pair(): st["p"] = ops.xattn_sublayers(x, [xs, xs], rows_per_kv=T * hw, lk=LK, scale=scale)
    def block(): st["b"] = ops.block_attn_sublayers(x, [xs, xs], tp, n_batch=nb, t_len=T, hw=hw, lk=LK, cross_scale=scale, temporal_scale=scale)

    I = 2048
    wu = (torch.randn(2 * I, C, generator=g) * C ** -0.5).half().float().to(dev); wd = (torch.randn(C, I, generator=g) * I ** -0.5).half().float().to(dev)
    bu = (torch.randn(2 * I, generator=g) * 0.2).to(dev); bd = (torch.randn(C, generator=g) * 0.1).to(dev)
    ffp = (gamma, beta, 1e-5, ops.pack_ff_weights(wu, wd, dev), bu, bd)
    def ff(): st["ff"] = ops.ff_sublayer(st["b"], *ffp)
    def whole(): st["w"] = ops.block_sublayers(x, [xs, xs], tp, ffp, n_batch=nb, t_len=T, hw=hw, lk=LK, cross_scale=scale, temporal_scale=scale)
    def whole_hilo(): st["wh"] = ops.block_sublayers(x, [xs, xs], tp, ffp, n_batch=nb, t_len=T, hw=hw, lk=LK, cross_scale=scale, temporal_scale=scale,
                                                     out_f32=False, out_hilo=True)

    fns = {"fused": fused, "cross_pair": pair, "block_of_three": block, "feed_forward": ff, "whole_block": whole, "whole_block_hilo_out": whole_hilo, "chain": chain, "layernorm": ln, "qkv": qkv, "temporal_attention": att, "to_out": out}
    chain(); fused(); pair(); block(); ff(); whole(); whole_hilo(); chain(); fused()
    torch.cuda.synchronize()
    t = {kk: [] for kk in fns}
    for _ in range(ROUNDS):
        for kk, f in fns.items():
            t[kk].append(time_once(f, PER))
    fl = 2.0 * m * 4 * C * C + 4.0 * m * T * C
    d = {"case": name, "rows": m, "max_abs_diff_fused_vs_chain": float((st["f"] - st["y"]).abs().max())}
    for kk in fns:
        d[kk + "_ms"] = {"median": round(statistics.median(t[kk]), 4), "min": round(min(t[kk]), 4)}
    med = d["fused_ms"]["median"]
    d["fused_tflops"] = round(fl / med / 1e9, 1)
    d["fused_algorithmic_GBps"] = round(8.0 * m * C / med / 1e6, 0)
    d["speedup_vs_chain"] = round(d["chain_ms"]["median"] / med, 3)
    print(json.dumps(d), flush=True)


def fcase(name, m):
    """The feed-forward sub-layer: fused launch against LayerNorm -> 512 -> 4096 GEGLU GEMM -> 2048 -> 512 GEMM + residual."""
    I = 2048
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(m, C, generator=g) * 1.3 + 0.2).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1).to(dev); beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    wu = (torch.randn(2 * I, C, generator=g) * C ** -0.5).half().float().to(dev); wd = (torch.randn(C, I, generator=g) * I ** -0.5).half().float().to(dev)
    bu = (torch.randn(2 * I, generator=g) * 0.2).to(dev); bd = (torch.randn(C, generator=g) * 0.1).to(dev)
    cu, cd = ops.pack_conv(wu, bu, geglu=True, device=dev), ops.pack_conv(wd, bd, device=dev)
    wp = ops.pack_ff_weights(wu, wd, dev)
    st = {}

    def ln(): st["n"] = ops.layernorm(x, gamma, beta, 1e-5)
    def up(): st["h"] = ops.linear(st["n"], cu)
    def down(): st["y"] = ops.linear(st["h"], cd, residual=x, out_f32=True)
    def down_hilo(): st["yh"] = ops.linear(st["h"], cd, residual=x, out_f32=True, out_hilo=True)
    def chain(): ln(); up(); down()
    def fused(): st["f"] = ops.ff_sublayer(x, gamma, beta, 1e-5, wp, bu, bd)
    def fused_hilo(): st["fh"] = ops.ff_sublayer(x, gamma, beta, 1e-5, wp, bu, bd, out_f32=False, out_hilo=True)

    fns = {"fused": fused, "fused_hilo_out": fused_hilo, "chain": chain, "layernorm": ln, "up_geglu": up, "down": down, "down_hilo_out": down_hilo}
    chain(); fused(); fused_hilo(); down_hilo(); chain(); fused()
    torch.cuda.synchronize()
    t = {kk: [] for kk in fns}
    for _ in range(ROUNDS):
        for kk, f in fns.items():
            t[kk].append(time_once(f, PER))
    fl = 2.0 * m * C * 3 * I
    d = {"case": name, "rows": m, "max_abs_diff_fused_vs_chain": float((st["f"] - st["y"]).abs().max()),
         "hilo_pair_bit_identical_to_chain": bool(torch.equal(st["fh"], ops.cast_hilo(st["f"])))}
    for kk in fns:
        d[kk + "_ms"] = {"median": round(statistics.median(t[kk]), 4), "min": round(min(t[kk]), 4)}
    med = d["fused_ms"]["median"]
    d["fused_tflops"] = round(fl / med / 1e9, 1)
    d["chain_tflops"] = round(fl / d["chain_ms"]["median"] / 1e9, 1)
    d["speedup_vs_chain"] = round(d["chain_ms"]["median"] / med, 3)
    print(json.dumps(d), flush=True)


if __name__ == "__main__":
    if "ff" in sys.argv[1:] or not sys.argv[1:]:
        fcase("feed-forward sub-layer, 160x160 level: 409 600 tokens", 2 * 8 * 160 * 160)
        fcase("feed-forward sub-layer, 80x80 level: 102 400 tokens", 2 * 8 * 80 * 80)
    if sys.argv[1:] == ["ff"]:
        sys.exit(0)
    if "temporal" in sys.argv[1:] or not sys.argv[1:]:
        tcase("temporal sub-layer, 160x160 level: 2 x 8 frames x 160 x 160 tokens", 2, 160, 160)
        tcase("temporal sub-layer, 80x80 level", 2, 80, 80)
    if sys.argv[1:] == ["temporal"]:
        sys.exit(0)
    case("160x160 level: 2 x 8 frames x 160 x 160 tokens", 2, 8 * 160 * 160)
    case("80x80 level: 2 x 8 frames x 80 x 80 tokens", 2, 8 * 80 * 80)
