"""Probe: D_s phase and D_t phase (two forwards + backward each) one after the other vs on two streams at once.
usage: python tools/d_overlap_probe.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import functional as Fn
from dvd_gan_amd.helpers import draw_frame_ids, sample_k_frames, vid_downsample
from dvd_gan_amd.train_step import Trainer

B = 64
cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                         total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9, n_class=101, k_sample=8)
dev = torch.device("cuda")
torch.manual_seed(0)
tr = Trainer([], cfg, device=dev)
real = (torch.rand(B, 48, 3, 64, 64) * 2 - 1).to(dev)
fake = (torch.rand(B, 48, 3, 64, 64) * 2 - 1).to(dev)
labels = torch.randint(0, 101, (B,)).to(dev)
ids = draw_frame_ids(48, 8)
real_s, fake_s = sample_k_frames(real, 48, 8, ids), sample_k_frames(fake, 48, 8, ids)
real_d, fake_d = vid_downsample(real), vid_downsample(fake)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def ds():
    return tr.calc_loss(tr.D_s(real_s, labels), True) + tr.calc_loss(tr.D_s(fake_s, labels), False)


def dt():
    return tr.calc_loss(tr.D_t(real_d, labels), True) + tr.calc_loss(tr.D_t(fake_d, labels), False)


def sequential():
    tr.reset_grad(); ds().backward(); Fn.join_side()
    tr.reset_grad(); dt().backward(); Fn.join_side()


def concurrent():
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    tr.reset_grad()
    with torch.cuda.stream(sA):
        la = ds()
    with torch.cuda.stream(sB):
        lb = dt()
    with torch.cuda.stream(sA):
        la.backward()
    with torch.cuda.stream(sB):
        lb.backward()
    cur.wait_stream(sA); cur.wait_stream(sB); Fn.join_side()


for name, fn in (("sequential", sequential), ("concurrent", concurrent), ("sequential", sequential), ("concurrent", concurrent)):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 3:.2f} ms per (D_s phase + D_t phase)")
