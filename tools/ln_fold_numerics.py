"""Offline numerics of folding LayerNorm into the consuming projection (next lever (ii) of DESIGN.md section 6; CPU only).

current path : y = fp16(LN(x)) . W^T            (LayerNorm output rounded to fp16, fp16 x fp16 products, fp32 accumulate)
folded       : y = rstd * (x . W'^T - mu * colsum(W')) + (W beta + b),  W' = fp16(W * gamma)
Both against the fp64 result, for rows with per-channel offsets and with a large common row mean (the cancellation case).
Result (this container): 2.0e-4 vs 2.0e-4 relative L2 in every case — the folded form loses nothing."""
import torch

torch.manual_seed(0)


def run(c, n, chan_scale, common, rows=4096):
    x = (torch.randn(rows, c, dtype=torch.float64) * (1 + torch.rand(1, c, dtype=torch.float64) * 3)
         + chan_scale * torch.randn(1, c, dtype=torch.float64) + common).half()
    g = 1 + 0.1 * torch.randn(c, dtype=torch.float64)
    b = 0.1 * torch.randn(c, dtype=torch.float64)
    w = (torch.randn(n, c, dtype=torch.float64) * c ** -0.5).half()
    bias = torch.randn(n, dtype=torch.float64) * 0.1
    xd = x.double()
    mu = xd.mean(1, keepdim=True)
    rstd = (xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    exact = ((xd - mu) * rstd * g + b) @ w.double().t() + bias
    ln16 = ((xd - mu) * rstd * g + b).float().half()
    cur = (ln16.float() @ w.float().t()).double() + bias
    wg = (w.double() * g).half()
    acc = (x.float() @ wg.float().t()).double()
    fold = rstd * (acc - mu * wg.double().sum(1)) + (w.double() @ b + bias)
    rel = lambda a: ((a - exact).norm() / exact.norm()).item()
    return rel(cur), rel(fold)


for c, n in ((512, 512), (1024, 1024)):
    for chan_scale, common in ((0.0, 0.0), (5.0, 0.0), (20.0, 0.0), (0.0, 10.0), (0.0, 50.0)):
        cur, fold = run(c, n, chan_scale, common)
        print(f"C={c} N={n} channel offsets x{chan_scale:4.1f} common mean {common:5.1f}: current {cur:.2e}  folded {fold:.2e}")
