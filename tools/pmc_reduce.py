"""Reduce a rocprofv3 --pmc rocpd database to one JSON line per kernel-name pattern: average duration and the summed counters
per launch.  usage: pmc_reduce.py <results.db> <label> <kernel-like-pattern>"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
label, pat = sys.argv[2], sys.argv[3]
rows = list(cur.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (pat,)))
out = {"variant": label}
if rows:
    launches = rows[0][2]
    out["launches"] = launches
    for name, total, n in rows:
        out[name] = total / n
try:
    d = list(cur.execute("select avg(duration) from kernels where name like ?", (pat,)))
    if d and d[0][0]:
        out["avg_ns"] = d[0][0]
except Exception as e:  # noqa: BLE001 — view names differ between rocprofv3 builds
    out["avg_ns_error"] = str(e)
print(json.dumps(out))
