"""Independent units of ONE clip on several HIP streams of one GPU.

The hot path is a chain of dependent launches, but a DDIM step under classifier-free guidance holds two independent UNet
evaluations (the unconditional and the text-conditioned branch), a long clip holds several temporal windows, and the decode
is three or more independent 3-frame chunks (the same units the multi-GPU path deals over ranks).  Issued on
separate streams from one host thread they fill each other's tails — a 256x256-tile conv launch on the coarse UNet levels is
1.6 … 6.3 rounds of workgroups over 256 CUs — and let HBM-bound passes (GroupNorm apply, LayerNorm) of one unit run beside
the MFMA-bound convs of the other.  The host issues a whole UNet forward in ≈ 14 ms against ≈ 250 ms of GPU time, so one
thread keeps every stream fed.

Ordering rules that make this safe with torch's caching allocator (no `record_stream` needed):
  * every side stream waits for the caller's stream before its first unit (inputs are produced there);
  * the caller's stream waits for every side stream used before `map` returns (outputs are consumed there);
  * tensors allocated while a side stream is current live in that stream's pool and are only reused by later work on the
    SAME stream; inputs freed by the caller are reused on the caller's stream, which by then waits for the side streams;
  * shared lazily-built device objects (packed weights, tables, text K/V) carry an event of their build (`engine.publish`) that every user stream waits for on the device (`engine.acquire`) until it has completed; the host never blocks.
Results are bit-identical to the serial evaluation: the units and their kernels are the same, only their order in time changes.
"""
import torch


class StreamSet:
    def __init__(self, device, n):
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(n)))]

    def map(self, items, fn):
        """[fn(it) for it in items] with item k issued on side stream k % n; the results are ordered on the caller's stream."""
        items = list(items)
        if len(items) <= 1:
            return [fn(it) for it in items]
        cur = torch.cuda.current_stream(self.device)
        used, outs = [], []
        for k, it in enumerate(items):
            st = self.streams[k % len(self.streams)]
            if st not in used:
                st.wait_stream(cur)
                used.append(st)
            with torch.cuda.stream(st):
                outs.append(fn(it))
        for st in used:
            cur.wait_stream(st)
        return outs


_SETS = {}


def stream_set(device, n):
    """Process-wide StreamSet per (device, n, calling stream): two host threads driving two clips on their own streams
    (bench.py --clips-per-step) must not share side streams."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(n),
           torch.cuda.current_stream(device).cuda_stream)
    s = _SETS.get(key)
    if s is None:
        s = _SETS[key] = StreamSet(device, n)
    return s
