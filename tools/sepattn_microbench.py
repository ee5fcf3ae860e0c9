"""SeparableAttn (Module/Attention.py:8-111) at the generator's shape -- [B, 48, 32, 32, 128] after module 8 -- forward + backward,
HIP-event time per pass; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
usage: python tools/sepattn_microbench.py [batch] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd.attention3d import SeparableAttn


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = SeparableAttn(128).to(dev)
    for c in m.model:
        c.gamma.data.fill_(0.3)
    x = (torch.randn(B, 48, 32, 32, 128, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(B, 48, 32, 32, 128, device=dev).to(torch.bfloat16)
    for it in range(iters + 1):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = m.run(x)
        e[1].record()
        y.backward(gy)
        e[2].record()
        torch.cuda.synchronize()
        if it:
            print("forward %.2f ms  backward %.2f ms" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
        x.grad = None
        for p in m.parameters():
            p.grad = None


if __name__ == "__main__":
    main()
