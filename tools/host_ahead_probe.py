"""Does the host run ahead of the GPU?  Wall time at which each train_step() call RETURNS (no synchronisation in between)
against the time the GPU needs for all of them.  A host that returns every ~GPU-step-time is being synchronised somewhere
inside the step (a pageable host->device copy, an .item()).   usage: python tools/host_ahead_probe.py [batch] [steps]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd.train_step import Trainer


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
    t0 = time.perf_counter()
    rets = []
    for _ in range(steps):
        tr.train_step(real, labels)
        rets.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("host returned at (ms):", [round(1e3 * r, 1) for r in rets])
    print("GPU done at %.1f ms -> %.1f ms per step; host per step %.1f ms" % (1e3 * total, 1e3 * total / steps, 1e3 * rets[-1] / steps))


if __name__ == "__main__":
    main()
