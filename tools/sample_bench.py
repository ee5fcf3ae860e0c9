"""Sampling path (trainer.py:323-334) against the training forward of the generator at config C2 (B clips, T=48, 64x64,
ch=32, bf16): time (HIP events, median of 5) and peak memory above the model's own.
usage: python tools/sample_bench.py [B]"""
import argparse
import os
import sys

import torch

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
    z, c = torch.randn(B, 120, device=dev), torch.randint(0, 101, (B,), device=dev)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b)); del out
        ts.sort()
        return ts[2], (torch.cuda.max_memory_allocated() - base) / 2 ** 30
    t_train, m_train = timed(lambda: tr.G(z, c))
    t_samp, m_samp = timed(lambda: tr.sample(z, c))
    print(f"B={B}: training forward {t_train:.1f} ms, {m_train:.1f} GiB above the model;  sample() {t_samp:.1f} ms, "
          f"{m_samp:.1f} GiB  ->  {t_train / t_samp:.2f}x faster, {m_train / max(m_samp, 1e-9):.1f}x less memory")


if __name__ == "__main__":
    main()
