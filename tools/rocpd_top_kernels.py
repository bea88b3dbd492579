"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; its `top_kernels` view is the `--stats` kernel summary.
usage: rocpd_top_kernels.py <results.db> <out.csv> ["# header comment"]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
name = "top_kernels" if "top_kernels" in views else next(v for v in views if "top_kernels" in v)
rows = list(cur.execute(f"select * from {name}"))
cols = [d[0] for d in cur.description]
with open(sys.argv[2], "w") as fh:
    if len(sys.argv) > 3:
        fh.write(sys.argv[3].rstrip() + "\n")
    fh.write(",".join(cols) + "\n")
    for r in rows:
        fh.write(",".join(f'"{v}"' if isinstance(v, str) and "," in v else (f"{v:.3f}" if isinstance(v, float) else str(v)) for v in r) + "\n")
print(f"{len(rows)} kernels -> {sys.argv[2]}")
