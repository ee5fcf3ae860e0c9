"""Kernel time per step by category from a rocprofv3 kernel-stats csv (tools/rocpd_summary.py output).
usage: python tools/stats_categories.py <stats.csv> <traced steps>"""
import csv
import sys

CATS = [("family (forward / backward-data convolutions)", ("conv_igemm", "conv_halo", "conv_group", "conv_thin")),
        ("weight gradients", ("conv_wgrad", "wgrad_thin_kernel")),
        ("weight-gradient reduce", ("wgrad_row_reduce", "wgrad_reduce", "wgrad_row_bias_reduce", "wgrad_bias_reduce", "wgrad_thin_reduce", "wgrad_thin")),
        ("ConvGRU gate kernels", ("gru_",)),
        ("conditional batch norm", ("cbn_", "bn_stats", "bn_finalize")),
        ("attention", ("attn_", "sep_")),
        ("spectral norm + weight packs", ("sn_", "pack_weight", "fragment_major", "thin_image", "thin_out_image")),
        ("ATen / runtime helpers", ("at::", "__amd_rocclr", "elementwise", "vectorized"))]
path, steps = sys.argv[1], float(sys.argv[2])
tot = {c: [0.0, 0] for c, _ in CATS}
tot["everything else"] = [0.0, 0]
for r in csv.DictReader(open(path)):
    name = r["kernel"]
    for c, keys in CATS:
        if any(k in name for k in keys):
            break
    else:
        c = "everything else"
    tot[c][0] += float(r["total_us"]); tot[c][1] += int(r["calls"])
s = 0.0
for c, (us, n) in tot.items():
    s += us
    print(f"{c:50s} {us / steps / 1e3:8.2f} ms / step   {n / steps:8.1f} launches")
print(f"{'sum':50s} {s / steps / 1e3:8.2f} ms / step   {sum(v[1] for v in tot.values()) / steps:8.1f} launches")
