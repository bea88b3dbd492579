"""Reduce the rocprofv3 PMC pass of tools/pmc_traffic.sh to per-launch HBM-side bytes of the conv kernels.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): TCC_EA0_RDREQ counts fabric read requests; wide streaming reads
are 128-B requests (FETCH_SIZE tallies them at 64 B and reports half), 32-B requests are counted separately in
TCC_EA0_RDREQ_32B.  Writes: 64-B requests are TCC_EA0_WRREQ_64B, the remainder 32 B.  Infinity-Cache hits are included
(these are fabric-side counters), so this is an upper bound on DRAM traffic.
"""
import json
import os
import sqlite3
import sys

args = [a for a in sys.argv[1:]]
out_path = None
if "--out" in args:
    k = args.index("--out"); out_path = args[k + 1]; del args[k:k + 2]
if not args:
    sys.exit("usage: pmc_traffic.py [--out file.json] <rocpd.db> [<rocpd.db> ...]")
c, launches = {}, None
for path in args:                      # one database per PMC pass (the counters of all passes are united)
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like '%conv_gemm%' "
                            "and kernel_name not like '%f32%' group by counter_name"))
    for name, total, n in rows:
        c[name] = total
        if launches is not None and n != launches:
            sys.exit(f"{path}: {n} conv launches for {name}, {launches} in another pass — not the same workload")
        launches = n
if launches is None:
    sys.exit("no conv_gemm rows in the counter tables")
rd = (c["TCC_EA0_RDREQ_sum"] - c.get("TCC_EA0_RDREQ_32B_sum", 0)) * 128 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
wr = c.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (c["TCC_EA0_WRREQ_sum"] - c.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "upscale-a-video_amd"))
from uav import build as _build  # noqa: E402
out = {"kernel_sources_digest": _build.conv_kernel_digest(),     # bench.py replays this file only for the library built from these sources
       "command": "rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum -- "
                  "python bench.py --no-cpu-baseline --no-kernel-events --no-throughput-mode --warmup 0 --steps 1 (tools/pmc_traffic.sh; "
                  f"{len(args)} PMC pass(es))",
       "kernels": "conv_gemm_kernel<*>, conv_gemm256i_kernel<*>, conv_gemm256w_kernel<*> (all fp16 implicit-GEMM launches)", "launches": launches, "counters": c,
       "read_bytes_per_launch": rd / launches, "write_bytes_per_launch": wr / launches,
       "hbm_bytes_per_launch": (rd + wr) / launches}
json.dump(out, open(out_path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_conv_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
