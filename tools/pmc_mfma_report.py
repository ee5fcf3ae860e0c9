"""rocprofv3 PMC pass over tools/conv_microbench.py (counter list tools/pmc_mfma.txt, csv output with --kernel-trace) ->
profiles/rNN_pmc_conv.json: per conv launch shape the matrix-pipe occupancy and the clock the chip held.
usage: python tools/pmc_mfma_report.py <rocprof output dir> <out.json> "<command line that was profiled>"

Derivations (MI355X_MICROARCH.md, per-instruction constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per
v_mfma_f32_32x32x16_bf16), summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE counts shader-clock cycles while the
kernel ran and is reported SUMMED over the 8 XCDs (raw value / duration = 13-14 "GHz"), so it is divided by 8 first.
pipe_busy = MFMA_BUSY / (1024 * GUI); clock = GUI / kernel duration;
tflops = INSTS_MFMA * 32768 flop / duration; at_full_pipe = 1024 SIMDs * 1024 flop/cycle * clock."""
import collections
import csv
import glob
import json
import sys

src, out_path, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
dur = {}                                          # dispatch id -> (kernel, ns, grid)
for path in glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))
cnt = collections.defaultdict(dict)               # dispatch id -> counter -> value
for path in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
groups = collections.defaultdict(list)
for did, c in cnt.items():
    if did not in dur or "conv_" not in dur[did][0] or "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"] == 0:
        continue
    name = dur[did][0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    groups[(name, dur[did][2], int(c["SQ_INSTS_MFMA"]))].append((dur[did][1], c))
rows = []
for (name, grid, insts), lst in sorted(groups.items(), key=lambda kv: -kv[0][2]):
    n = len(lst)
    ns = sum(d for d, _ in lst) / n
    avg = lambda k: sum(c.get(k, 0.0) for _, c in lst) / n
    gui = avg("GRBM_GUI_ACTIVE") / 8.0            # per XCD
    busy = avg("SQ_VALU_MFMA_BUSY_CYCLES")
    clock = gui / ns if ns else 0.0               # GHz
    rows.append({"kernel": name, "grid_threads": grid, "launches": n, "duration_us": round(ns / 1e3, 1),
                 "SQ_INSTS_MFMA": insts, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_WAVE_CYCLES": avg("SQ_WAVE_CYCLES"),
                 "SQ_BUSY_CYCLES": avg("SQ_BUSY_CYCLES"), "SQ_WAIT_ANY": avg("SQ_WAIT_ANY"),
                 "SQ_WAIT_INST_ANY": avg("SQ_WAIT_INST_ANY"), "SQ_ACTIVE_INST_ANY": avg("SQ_ACTIVE_INST_ANY"),
                 "GRBM_GUI_ACTIVE_per_xcd": gui,
                 "derived": {"clock_ghz": round(clock, 3), "pipe_busy": round(busy / (1024 * gui), 4) if gui else None,
                             "tflops": round(insts * 32768 / ns / 1e3, 1) if ns else None,
                             "tflops_at_full_pipe_at_this_clock": round(1024 * 1024 * clock / 1e3, 1)}})
json.dump({"source": cmd, "note": "profiled passes clock lower than un-profiled runs (MI355X_MICROARCH.md, DVFS): compare ratios",
           "launch_shapes": rows}, open(out_path, "w"), indent=1)
for r in rows[:12]:
    print(r["kernel"][:60], r["grid_threads"], r["duration_us"], r["derived"])
