"""Discriminator phases of one step in isolation (config C2 shapes: 64 clips, k=8 sampled frames for D_s, 48 x 32 x 32 for D_t),
for `rocprofv3 --kernel-trace --stats`: D_s / D_t forward on real + fake, backward, and the frozen-weight pass of the generator
step.  usage: python tools/d_profile.py [iters]   (prints HIP-event times per phase)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import functional as Fn
from dvd_gan_amd.train_step import Trainer


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    B = 64
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tr = Trainer([], cfg, device=dev)
    labels = torch.randint(0, 101, (B,), device=dev)
    real_s = torch.rand(B, 8, 3, 64, 64, device=dev) * 2 - 1
    fake_s = (torch.rand(B, 8, 3, 64, 64, device=dev) * 2 - 1).requires_grad_(True)
    real_d = torch.rand(B, 3, 48, 32, 32, device=dev) * 2 - 1
    fake_d = (torch.rand(B, 3, 48, 32, 32, device=dev) * 2 - 1).requires_grad_(True)
    tot = {}

    def phase(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        tot[name] = tot.get(name, 0.0) + a.elapsed_time(b)
        return out
    for it in range(iters + 1):
        if it == 1:
            tot.clear()
        for name, net, opt, xr, xf in (("D_s", tr.D_s, tr.ds_optimizer, real_s, fake_s), ("D_t", tr.D_t, tr.dt_optimizer, real_d, fake_d)):
            loss = phase(name + " forward real+fake", lambda: tr.calc_loss(net(xr, labels), True) + tr.calc_loss(net(xf.detach(), labels), False))
            opt.zero_grad()
            phase(name + " backward", lambda: (loss.backward(), Fn.join_side()))
            phase(name + " adam", lambda: opt.step())
            tr._freeze_d(True)
            g = phase(name + " frozen forward (G step)", lambda: tr.calc_loss(net(xf, labels), True))
            phase(name + " frozen backward to the clips", lambda: (g.backward(), Fn.join_side()))
            tr._freeze_d(False)
    for n, v in tot.items():
        print(f"{n:40s} {v / iters:8.2f} ms")
    print(f"{'sum':40s} {sum(tot.values()) / iters:8.2f} ms")


if __name__ == "__main__":
    main()
