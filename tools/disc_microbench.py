"""The discriminator phases of a step alone (config C2): D_s and D_t over real + fake clips with their backward passes (the D step)
and over the fake clips with the gradient to the clips only (the generator step's passes).  HIP-event time per phase; under
`rocprofv3 --kernel-trace --stats` the per-kernel split.   usage: python tools/disc_microbench.py [batch] [iters]"""
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
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tr = Trainer([], cfg, device=dev)
    real = (torch.rand(B, 48, 3, 64, 64, device=dev) * 2 - 1)
    labels = torch.randint(0, 101, (B,)).to(dev)
    T, k = 48, 8
    for it in range(iters + 1):
        fake = (torch.rand(B, 48, 3, 64, 64, device=dev) * 2 - 1).requires_grad_(True)
        zc = tr.label_sample()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        real_s = sample_k_frames(real, T, k, draw_frame_ids(T, k))
        fake_s = sample_k_frames(fake, T, k, draw_frame_ids(T, k))
        l1 = tr.calc_loss(tr.D_s(real_s, labels), True) + tr.calc_loss(tr.D_s(fake_s.detach(), zc), False)
        tr.reset_grad(); l1.backward(); Fn.join_side()
        ev[1].record()
        real_d, fake_d = vid_downsample(real), vid_downsample(fake)
        l2 = tr.calc_loss(tr.D_t(real_d, labels), True) + tr.calc_loss(tr.D_t(fake_d.detach(), zc), False)
        tr.dt_optimizer.zero_grad(); l2.backward(); Fn.join_side()
        ev[2].record()
        tr._freeze_d(True)
        l3 = tr.calc_loss(tr.D_s(fake_s, zc), True) + tr.calc_loss(tr.D_t(fake_d, zc), True)
        tr._freeze_d(False)
        l3.backward(); Fn.join_side()
        ev[3].record()
        torch.cuda.synchronize()
        if it:
            print("D_s step %.2f ms   D_t step %.2f ms   generator-step passes %.2f ms" %
                  (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])))


if __name__ == "__main__":
    main()
