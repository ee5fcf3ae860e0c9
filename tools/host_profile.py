"""Where the HOST time of a training step goes: cProfile over a few steps at a batch small enough that the GPU never makes the host
wait (B=8: the launch sequence is the one of B=64).   usage: python tools/host_profile.py [batch] [steps] [rows]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd.train_step import Trainer


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tr = Trainer([], cfg, device=dev)
    real = (torch.rand(B, 3, 48, 64, 64) * 2 - 1).to(dev)
    labels = torch.randint(0, 101, (B,)).to(dev)
    tr.register_label_buffer(labels)
    for _ in range(3):
        tr.train_step(real, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(real, labels)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("B=%d: host %.1f ms per step, GPU done after %.1f ms per step" % (B, 1e3 * host / steps, 1e3 * total / steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        tr.train_step(real, labels)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(rows)
    st.sort_stats("cumulative").print_stats(rows)


if __name__ == "__main__":
    main()
