"""Phase trace of the fused cross-attention sub-layer kernel (csrc/xattn_fused.hip): runs the s_memtime-stamped development instance
(tools/ab/libuav_xattn_dev.so, built by tools/ab/build_dev.sh — not part of libuav_hip.so) on the two 512-channel levels and prints the
mean number of shader-clock ticks every workgroup spends per phase, plus the span of the whole launch.
usage: python tools/trace_xattn.py"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "upscale-a-video_amd"))
from uav import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(ROOT, "tools", "ab", "libuav_xattn_dev.so"))
c_p, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
lib.uav_dev_xattn_sublayers_trace.restype = C.c_int
lib.uav_dev_xattn_sublayers_trace.argtypes = [c_p, c_p, c_p, i32, i64, i32, i32, f32, c_p, c_p]
from uav._lib import XattnParams  # noqa: E402
PHASES = ["statistics pass (1st read of x)", "operands + accumulators (2nd read)", "head 0 (cold ring)", "head 1: Q GEMM, 64 MFMA",
          "head 1: S = K Q, 12 MFMA", "head 1: softmax", "head 1: O = V P, 12 MFMA", "head 1: acc += Wout O, 64 MFMA", "heads 2 .. 7",
          "drain + stores"]


def case(name, nb, rpk, lk=77):
    Cc, D = 512, 64
    m = nb * rpk
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(m, Cc, generator=g) * 1.3 + 0.2).to(dev)
    gamma = (torch.randn(Cc, generator=g) * 0.2 + 1).to(dev); beta = (torch.randn(Cc, generator=g) * 0.1).to(dev)
    wq = (torch.randn(Cc, Cc, generator=g) * Cc ** -0.5).half().float().to(dev); wo = (torch.randn(Cc, Cc, generator=g) * Cc ** -0.5).half().float().to(dev)
    bo = (torch.randn(Cc, generator=g) * 0.1).to(dev)
    kv = torch.randn(nb * lk, 2 * Cc, generator=g).half().to(dev)
    wqp, wop = ops.pack_xattn_weight(wq, "q", dev), ops.pack_xattn_weight(wo, "out", dev)
    kvp = ops.xattn_pack_kv(kv[:, :Cc], kv[:, Cc:], n_batch=nb, lk=lk, k_stride=2 * Cc, v_stride=2 * Cc)
    ref = ops.xattn_sublayer(x, gamma, beta, 1e-5, wqp, kvp, wop, bo, rows_per_kv=rpk, lk=lk, scale=D ** -0.5)
    out = torch.empty_like(x)
    ntile = m // 128
    tr = torch.zeros(ntile * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    arr = (XattnParams * 1)()
    arr[0].ln_gamma, arr[0].ln_beta, arr[0].ln_eps = gamma.data_ptr(), beta.data_ptr(), 1e-5
    arr[0].wq_packed, arr[0].kv_packed, arr[0].wo_packed, arr[0].out_bias = wqp.data_ptr(), kvp.data_ptr(), wop.data_ptr(), bo.data_ptr()
    for _ in range(3):
        rc = lib.uav_dev_xattn_sublayers_trace(x.data_ptr(), out.data_ptr(), C.cast(arr, c_p), 1, m, rpk, lk, D ** -0.5, tr.data_ptr(), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "the stamped instance must compute what the product instance computes"
    t = tr.reshape(ntile, 16).cpu().double()
    d = {"case": name, "workgroups": ntile}
    for i, ph in enumerate(PHASES):
        d[ph] = round(float((t[:, i + 1] - t[:, i]).mean()), 0)
    d["whole workgroup (mean ticks)"] = round(float((t[:, 10] - t[:, 0]).mean()), 0)
    d["whole launch (ticks, first start to last end)"] = float(t[:, 10].max() - t[:, 0].min())
    print(json.dumps(d), flush=True)


if __name__ == "__main__":
    case("160x160 level, M = 409600", 2, 8 * 160 * 160)
    case("80x80 level, M = 102400", 2, 8 * 80 * 80)
    case("3 waves of tiles, M = 98304", 1, 98304)
