"""Timeline of the LAST step in a tools/timeline_dump.py CSV: how much of the step the GPU runs one stream, both streams,
nothing, and how much of the main-stream time is spent in launches too small to fill the chip.
usage: python tools/timeline_report.py <timeline.csv> [steps_in_trace]"""
import csv
import collections
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["start_ns"]), int(r["end_ns"]), r["stream"], int(r["workgroups"]), r["name"]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# the last step: the dispatches after the (nsteps-1)/nsteps point of the count (steps launch identical sequences)
per = len(rows) // nsteps
rows = rows[len(rows) - per:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = (t1 - t0) / 1e6
streams = collections.Counter()
for s, e, st, wg, n in rows:
    streams[st] += (e - s) / 1e6
main = max(streams, key=streams.get)
print("span %.1f ms; kernel time by stream (ms):" % span, {k: round(v, 1) for k, v in streams.items()}, "main =", main)
# sweep
ev = []
for s, e, st, wg, n in rows:
    cls = "M" if st == main else "S"
    small = cls == "M" and wg < 512
    ev.append((s, 1, cls, small)); ev.append((e, -1, cls, small))
ev.sort()
cnt = {"M": 0, "S": 0, "small": 0}
acc = collections.Counter()
last = t0
for t, d, cls, small in ev:
    dt = (t - last) / 1e6
    if cnt["M"] and cnt["S"]:
        key = "both"
    elif cnt["M"]:
        key = "main only"
    elif cnt["S"]:
        key = "side only"
    else:
        key = "idle"
    acc[key] += dt
    if cnt["M"] and cnt["small"] == cnt["M"]:
        acc["main small (<512 wg)" + (" + side" if cnt["S"] else " alone")] += dt
    last = t
    cnt[cls] += d
    if small:
        cnt["small"] += d
for k, v in sorted(acc.items()):
    print("  %-32s %8.1f ms" % (k, v))
# small main-stream launches by kernel
sm = collections.defaultdict(lambda: [0, 0.0])
for s, e, st, wg, n in rows:
    if st == main and wg < 512:
        k = n.split("(")[0][:70]
        sm[k][0] += 1; sm[k][1] += (e - s) / 1e6
print("small main-stream launches:")
for k, v in sorted(sm.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %6d %8.2f ms  %s" % (v[0], v[1], k))
# idle time by what runs around it: the kernel that ended before a gap of the main stream (host-bound stretches show up as many
# gaps after tiny kernels)
gaps = collections.defaultdict(lambda: [0, 0.0])
mrows = sorted(r for r in rows)
busy_until = mrows[0][1]
prev = mrows[0][4]
for s, e, st, wg, n in mrows[1:]:
    if s > busy_until:
        k = prev.split("(")[0][:60]
        gaps[k][0] += 1; gaps[k][1] += (s - busy_until) / 1e6
    if e > busy_until:
        busy_until, prev = e, n
print("GPU-idle gaps (no kernel on any stream), by the kernel that ran before the gap:")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %6d %8.2f ms  %s" % (v[0], v[1], k))
