"""rocprofv3 (rocpd sqlite output) -> per-kernel stats CSV, the `--stats` summary in a diffable form.
usage: python tools/rocpd_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, round(tot, 1), round(avg, 2), round(pct, 3)])
print("kernels:", len(rows), "total ms:", round(sum(r[2] for r in rows) / 1e3, 1))
