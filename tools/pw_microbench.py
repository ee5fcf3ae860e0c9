"""Times the HBM-bound kernels of the step (BN statistics, CBN apply / backward) on config-C2 shapes and
prints achieved GB/s against the algorithmic bytes (HIP events).  usage: python tools/pw_microbench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import kern as K

SHAPES = [(3072, 64, 64, 64), (3072, 32, 32, 128), (3072, 16, 16, 256), (3072, 8, 8, 512)]   # frames, H, W, C


def bench(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev, dt, B = "cuda", torch.bfloat16, 64
    for F_, H, W, C in SHAPES:
        x = torch.randn(F_, H, W, C, device=dev).to(dt)
        g = torch.randn(F_, H, W, C, device=dev).to(dt)
        gb = torch.randn(B, 2 * C, device=dev)
        samp = (torch.arange(F_, device=dev, dtype=torch.int32) % B).contiguous()
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbytes = x.numel() * 2
        ms = bench(lambda: K.bn_stats(x, C, True, 1e-5, 0.1, rm, rv), iters)
        print(f"bn_stats     {F_}x{H}x{W}x{C}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s (1 pass)", flush=True)
        mean, rstd = K.bn_stats(x, C, True, 1e-5, 0.1, rm, rv)
        ms = bench(lambda: K.cbn_apply(x, C, mean, rstd, gb, samp, True), iters)
        print(f"cbn_apply    {F_}x{H}x{W}x{C}: {ms * 1e3:8.1f} us  {2 * nbytes / ms / 1e6:7.0f} GB/s (2 passes)", flush=True)
        a = K.cbn_apply(x, C, mean, rstd, gb, samp, True)
        ms = bench(lambda: K.cbn_backward(g, a, x, C, mean, rstd, gb, samp, True), iters)
        print(f"cbn_backward {F_}x{H}x{W}x{C}: {ms * 1e3:8.1f} us  {5 * nbytes / ms / 1e6:7.0f} GB/s (2+3 passes)", flush=True)


if __name__ == "__main__":
    main()
