"""Where do the ATen / runtime helper launches of a training step come from?  One step under torch.profiler (with Python stacks);
every aten fill / zero / copy / add / cat ... is attributed to the innermost dvd_gan_amd source line that issued it.
usage: python tools/small_ops_report.py [batch]   -> table: count per step, device time, op, source line"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd.train_step import Trainer


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tr = Trainer([], cfg, device=dev)
    real = (torch.rand(B, 3, 48, 64, 64) * 2 - 1).to(dev)
    labels = torch.randint(0, 101, (B,)).to(dev)
    tr.register_label_buffer(labels)
    for _ in range(2):
        tr.train_step(real, labels)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.train_step(real, labels)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
            continue
        if not ev.kernels:                       # only the op that launched the kernel itself (not its aten:: parents)
            continue
        site = "?"
        for fr in ev.stack or []:
            if "dvd_gan_amd" in fr or "bench.py" in fr:
                site = fr.strip().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "")
                break
        k = (ev.name, site)
        agg[k][0] += 1
        agg[k][1] += sum(kk.duration for kk in ev.kernels)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    tot_n = sum(v[0] for v in agg.values())
    tot_t = sum(v[1] for v in agg.values())
    print(f"ATen-launched kernels in one step: {tot_n}, {tot_t / 1e3:.2f} ms of device time")
    for (name, site), (n, t) in rows[:60]:
        print(f"{n:5d} {t / 1e3:8.3f} ms  {name:28s} {site}")


if __name__ == "__main__":
    main()
