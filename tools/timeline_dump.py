"""rocprofv3 (rocpd sqlite) -> one CSV row per kernel dispatch: start_ns, end_ns, stream, queue, workgroups, name.
usage: python tools/timeline_dump.py <results.db> <out.csv>      (input of tools/timeline_report.py)"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in names else None
if view is None:
    print("no `kernels` view; objects:", names)
    sys.exit(1)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)


def pick(*cands):
    for c in cands:
        if c in cols:
            return c
    return None


c_start, c_end, c_name = pick("start"), pick("end"), pick("name", "kernel_name")
c_stream, c_queue = pick("stream_id", "stream"), pick("queue_id", "queue")
gx, gy, gz = pick("grid_x", "grid_size_x"), pick("grid_y", "grid_size_y"), pick("grid_z", "grid_size_z")
wx, wy, wz = pick("workgroup_x", "workgroup_size_x"), pick("workgroup_y", "workgroup_size_y"), pick("workgroup_z", "workgroup_size_z")
sel = [c_start, c_end, c_stream or "0", c_queue or "0", gx or "0", gy or "1", gz or "1", wx or "1", wy or "1", wz or "1", c_name]
rows = list(db.execute("select %s from kernels order by %s" % (", ".join(sel), c_start)))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["start_ns", "end_ns", "stream", "queue", "workgroups", "name"])
    for s, e, st, q, a, b, c, x, y, z, n in rows:
        wg = (int(a) // max(int(x), 1)) * (int(b) // max(int(y), 1)) * (int(c) // max(int(z), 1))
        w.writerow([s, e, st, q, wg, n[:90]])
print("dispatches:", len(rows))
