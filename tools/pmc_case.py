"""Per-dispatch fabric-side bytes of the conv launches of a micro-benchmark run under `rocprofv3 --pmc TCC_EA0_*` (same counters and
gfx950 corrections as tools/pmc_traffic.py).  usage: pmc_case.py <results.db> [launches-per-case]
Prints one line per group of consecutive conv dispatches (a case of tools/bench_epilogue.py = warm-up + timed launches)."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
per_case = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
key = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else None)
if key is None:
    print(json.dumps({"error": "no dispatch key", "columns": cols}))
    sys.exit(0)
rows = list(cur.execute(f"select {key}, kernel_name, counter_name, sum(value) from counters_collection where kernel_name like '%conv_gemm%' "
                        f"group by {key}, kernel_name, counter_name order by {key}"))
disp = {}
for d, k, c, v in rows:
    disp.setdefault(d, {"kernel": k})[c] = v
order = sorted(disp)
for i in range(0, len(order), per_case):
    grp = [disp[d] for d in order[i:i + per_case]]
    n = len(grp)
    c = {name: sum(g.get(name, 0) for g in grp) / n for name in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")}
    rd = (c["TCC_EA0_RDREQ_sum"] - c["TCC_EA0_RDREQ_32B_sum"]) * 128 + c["TCC_EA0_RDREQ_32B_sum"] * 32
    wr = c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32
    kern = grp[0]["kernel"]
    kern = kern[kern.find("conv_gemm"):][:40]
    print(json.dumps({"case_index": i // per_case, "kernel": kern, "launches": n, "read_MB": round(rd / 1e6, 1), "write_MB": round(wr / 1e6, 1),
                      "rd_32B_share": round(c["TCC_EA0_RDREQ_32B_sum"] / max(1, c["TCC_EA0_RDREQ_sum"]), 3),
                      "wr_64B_share": round(c["TCC_EA0_WRREQ_64B_sum"] / max(1, c["TCC_EA0_WRREQ_sum"]), 3)}))
