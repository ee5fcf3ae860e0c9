"""Where one training step spends its time, by phase (HIP events on the step's stream; config C2 unless overridden).
usage: python tools/phase_times.py [batch] [steps]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import functional as Fn
from dvd_gan_amd.helpers import draw_frame_ids, sample_k_frames, vid_downsample
from dvd_gan_amd.train_step import Trainer


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tr = Trainer([], cfg, device=dev)
    real = (torch.rand(B, 3, 48, 64, 64) * 2 - 1).to(dev)
    labels = torch.randint(0, 101, (B,)).to(dev)
    tr.train_step(real, labels)
    torch.cuda.synchronize()
    names, marks = [], []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        names.append(name); marks.append(e)
    tot = {}
    for _ in range(steps):
        names.clear(); marks.clear()
        T, k = 48, 8
        mark("start")
        rv = real.permute(0, 2, 1, 3, 4).contiguous()
        real_s = sample_k_frames(rv, T, k, draw_frame_ids(T, k))
        z, zc = torch.randn(B, 120).to(dev), tr.label_sample()
        mark("prep")
        fake = tr.G(z, zc)
        mark("G forward")
        fake_s = sample_k_frames(fake, T, k, draw_frame_ids(T, k))
        l1 = tr.calc_loss(tr.D_s(real_s, labels), True) + tr.calc_loss(tr.D_s(fake_s.detach(), zc), False)
        mark("D_s forward x2")
        tr.reset_grad(); l1.backward(); Fn.join_side()
        mark("D_s backward")
        tr.ds_optimizer.step()
        real_d, fake_d = vid_downsample(rv), vid_downsample(fake)
        l2 = tr.calc_loss(tr.D_t(real_d, labels), True) + tr.calc_loss(tr.D_t(fake_d.detach(), zc), False)
        mark("D_t forward x2 (+ D_s Adam, downsample)")
        tr.dt_optimizer.zero_grad(); l2.backward(); Fn.join_side()
        mark("D_t backward")
        tr.dt_optimizer.step()
        tr._freeze_d(True)
        l3 = tr.calc_loss(tr.D_s(fake_s, zc), True) + tr.calc_loss(tr.D_t(fake_d, zc), True)
        tr._freeze_d(False)
        mark("G-step D forwards")
        tr.g_optimizer.zero_grad(); l3.backward(); Fn.join_side()
        mark("G-step backward (D_s, D_t data grads + G)")
        tr.g_optimizer.step()
        mark("G Adam")
        torch.cuda.synchronize()
        for i in range(1, len(names)):
            tot[names[i]] = tot.get(names[i], 0.0) + marks[i - 1].elapsed_time(marks[i])
    s = sum(tot.values()) / steps
    for n, v in tot.items():
        print(f"{n:48s} {v / steps:8.2f} ms  {100 * v / steps / s:5.1f} %")
    print(f"{'step':48s} {s:8.2f} ms")


if __name__ == "__main__":
    main()
