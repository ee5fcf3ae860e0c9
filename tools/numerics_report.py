"""bf16 production mode vs exact f32-MFMA mode of the SAME HIP path at the real architecture size
(ch=32, T=48, 64x64, 101 classes), B=4: generator output, discriminator outputs, gradient cosines.
(The exact mode itself is pinned to the reference goldens by tests/test_gpu_*.py.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd.gen_net import Generator
from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
from dvd_gan_amd.helpers import sample_k_frames, vid_downsample


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def main():
    dev = "cuda"
    B, T, ch, ncls = 4, 48, 32, 101
    torch.manual_seed(0)
    z = torch.randn(B, 120, device=dev)
    cls = torch.randint(0, ncls, (B,), device=dev)
    ids = torch.arange(0, T, 6)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(1)
        G = Generator(120, 4, ncls, ch, T, compute_dtype=dt).to(dev).train()
        Ds = SpatialDiscriminator(ch, ncls, compute_dtype=dt).to(dev)
        Dt = TemporalDiscriminator(ch, ncls, compute_dtype=dt).to(dev)
        Ds.attn.gamma.data.fill_(0.3)
        Dt.self_attn.gamma.data.fill_(0.3)
        fake = G(z, cls)
        ds = Ds(sample_k_frames(fake, T, 8, ids), cls)
        dtt = Dt(vid_downsample(fake), cls)
        loss = torch.relu(1 - ds).mean() + torch.relu(1 - dtt).mean()
        loss.backward()
        res[dt] = dict(fake=fake.detach(), ds=ds.detach(), dt=dtt.detach(), loss=float(loss),
                       grads={k: p.grad.detach().clone() for k, p in G.named_parameters() if p.grad is not None})
    a, b = res[torch.float32], res[torch.bfloat16]
    print(f"generator output  rel-L2 bf16 vs f32 : {rel(b['fake'], a['fake']):.3e}")
    print(f"D_s output        rel-L2             : {rel(b['ds'], a['ds']):.3e}")
    print(f"D_t output        rel-L2             : {rel(b['dt'], a['dt']):.3e}")
    print(f"loss f32 {a['loss']:.5f}  bf16 {b['loss']:.5f}")
    cs = sorted((cos(b['grads'][k], a['grads'][k]), k) for k in a['grads'] if a['grads'][k].abs().max() > 0)
    print(f"G gradient cosine: min {cs[0][0]:.4f} ({cs[0][1]}), median {cs[len(cs) // 2][0]:.4f}, "
          f"5th pct {cs[len(cs) // 20][0]:.4f} over {len(cs)} tensors")
    big = [c for c, k in cs if a['grads'][k].numel() > 10000]
    print(f"  weight tensors > 10k elements: min cosine {min(big):.4f}")


if __name__ == "__main__":
    main()
