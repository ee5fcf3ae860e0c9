"""rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) -> profiles/rNN_hbm_traffic.json.
usage: python tools/traffic_summary.py <pmc_dir> <out.json> "<build note>"
Read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section: gfx950 counts 128-B requests of wide reads as
64 B); WRITE_SIZE is used as reported.  Both counters are in KiB."""
import collections
import csv
import glob
import json
import sys

pmc_dir, out_path, note = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(float)
cnt = collections.Counter()
per = collections.defaultdict(lambda: [0.0, 0.0, 0])        # kernel name -> [fetch KiB, write KiB, launches]
for path in sorted(glob.glob(f"{pmc_dir}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"]
        # forward / backward-data family = halo-staged + tap-by-tap kernels; weight-gradient family = row + tap kernels
        k = "conv_igemm" if ("conv_igemm" in kn or "conv_halo" in kn or "conv_group" in kn or "conv_thin" in kn) else "conv_wgrad" if ("conv_wgrad" in kn or "wgrad_thin" in kn) else "other"
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
        short = kn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        e = per[short]
        if r["Counter_Name"] == "FETCH_SIZE":
            e[0] += float(r["Counter_Value"]); e[2] += 1
        elif r["Counter_Name"] == "WRITE_SIZE":
            e[1] += float(r["Counter_Value"])
out = {"source": "rocprofv3 -i tools/pmc_traffic.txt (separate passes: FETCH_SIZE, then WRITE_SIZE) --kernel-trace -- python "
                 "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-prof; config C2, B=64, bf16; " + note,
       "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB",
       "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) "
                     "reads -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE uncorrected (uncalibrated)",
       "shape": {"size": 64, "frames": 48, "batch": 64, "ch": 32, "dtype": "bf16"},      # bench.py reports `traffic` only for it
       "kernels": {}}
for k in ("conv_igemm", "conv_wgrad", "other"):
    f, nf = agg[(k, "FETCH_SIZE")], cnt[(k, "FETCH_SIZE")]
    w, nw = agg[(k, "WRITE_SIZE")], cnt[(k, "WRITE_SIZE")]
    if not nf or not nw:
        continue
    out["kernels"][k] = {"launches_per_step": nf, "fetch_kib_per_launch_raw": round(f / nf, 1),
                         "write_kib_per_launch": round(w / nw, 1),
                         "hbm_bytes_per_launch_corrected": int((2 * f / nf + w / nw) * 1024),
                         "hbm_gb_per_step_corrected": round((2 * f + w) * 1024 / 1e9, 1)}
top = sorted(per.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:16]
out["by_kernel_gb_per_step"] = [{"kernel": k[:90], "launches": v[2], "read_gb": round(2 * v[0] * 1024 / 1e9, 1),
                                 "write_gb": round(v[1] * 1024 / 1e9, 1)} for k, v in top]
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
