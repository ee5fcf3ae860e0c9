"""Aggregate the per-launch CSV written by libdvdgan_hip (DVD_PROF_CSV) by kernel kind + shape."""
import collections
import sys

rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(sys.argv[1]):
    f = line.strip().split(",")
    kind, M, C, Cout, taps, split, flags, ms, flops = f[:9]
    var = int(f[9]) if len(f) > 9 else 0          # kernel variant (see dvd_prof_report_variants)
    key = (int(kind), int(M), int(C), int(Cout), int(taps), int(split), int(flags), var)
    r = rows[key]
    r[0] += 1
    r[1] += float(ms)
    r[2] += float(flops)
tot = sum(r[1] for r in rows.values())
print(f"{'kind':>4} {'M':>9} {'C':>5} {'Cout':>5} {'taps':>4} {'split':>5} {'flg':>3} {'var':>3} {'n':>5} {'ms':>9} {'%':>5} {'TF/s':>7} {'us/launch':>9}")
for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{key[0]:>4} {key[1]:>9} {key[2]:>5} {key[3]:>5} {key[4]:>4} {key[5]:>5} {key[6]:>3} {key[7]:>3} {r[0]:>5} {r[1]:>9.2f} {100 * r[1] / tot:>5.1f} "
          f"{r[2] / (r[1] * 1e-3) / 1e12 if r[1] else 0:>7.1f} {1e3 * r[1] / r[0]:>9.1f}")
print("total ms", round(tot, 1), " by kind:", {k: round(sum(r[1] for kk, r in rows.items() if kk[0] == k), 1) for k in sorted({kk[0] for kk in rows})})
