"""Race screen for the weights-from-L2 convolution kernels: every variant (256 x 128, 128 x 128, thin; ReLU-on-load, nearest x2,
split-K slabs, 3-D taps) against the LDS-staged kernel on the same operands, many repetitions, bitwise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import kern as K

CASES = [  # frames, S, Cin, Cout, k, up2, relu, nsplit
    (64, 32, 512, 256, 5, 0, 0, 1), (64, 16, 1024, 512, 5, 0, 0, 2), (256, 32, 256, 384, 5, 0, 0, 1),
    (64, 32, 128, 128, 3, 0, 0, 1), (64, 16, 256, 256, 3, 0, 0, 1), (64, 16, 256, 256, 3, 0, 0, 2),
    (512, 64, 64, 64, 3, 0, 0, 1), (512, 64, 8, 64, 3, 0, 0, 1), (256, 32, 256, 128, 3, 1, 1, 1), (512, 64, 128, 64, 3, 1, 1, 1),
    (256, 32, 128, 128, 3, 0, 1, 1), (64, 32, 256, 512, 5, 0, 0, 1), (3072, 16, 256, 256, 3, 0, 0, 1),
    # frames of 8 x 8 / 4 x 4 pixels (whole-frame footprints): split and unsplit, 256- and 128-row tiles, ragged frame counts
    (64, 8, 512, 1024, 5, 0, 0, 4), (64, 8, 512, 512, 5, 0, 0, 8), (64, 8, 256, 512, 3, 0, 0, 4), (64, 8, 256, 256, 3, 0, 0, 1),
    (3072, 8, 256, 1536, 5, 0, 0, 1), (3072, 8, 256, 256, 3, 0, 1, 1), (64, 4, 512, 1024, 5, 0, 0, 8), (64, 4, 256, 512, 3, 0, 0, 8),
    (3072, 4, 256, 768, 3, 0, 0, 1), (61, 8, 64, 96, 3, 0, 0, 1), (13, 4, 40, 72, 5, 0, 1, 2),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev, dt = "cuda", torch.bfloat16
    bad = 0
    for F_, S, Cin, Cout, k, up2, relu, ns in CASES:
        Sin = S // 2 if up2 else S
        x = torch.randn(F_, Sin, Sin, Cin, device=dev).to(dt)
        pk = K.PackedConv(dt, Cout, Cin, (k, k), dev).fill(torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
        wq = pk.fragment_major("wf")
        kw = dict(up2=bool(up2), relu_in=bool(relu), nsplit=ns, slabs=ns > 1)
        ref = K.conv_forward(x, pk.wf, (k, k), Cout, **kw).clone()
        n_bad = 0
        for _ in range(reps):
            got = K.conv_forward(x, pk.wf, (k, k), Cout, wq=wq, **kw)
            if not torch.equal(ref, got):
                n_bad += 1
        bad += n_bad
        print(f"F={F_} S={S} C={Cin}->{Cout} k={k} up2={up2} relu={relu} ns={ns}: {n_bad}/{reps} runs differ", flush=True)
    # 3-D taps (D_t): [B, T, H, W, C]
    x = torch.randn(16, 12, 32, 32, 64, device=dev).to(dt)
    pk = K.PackedConv(dt, 64, 64, (3, 3, 3), dev).fill(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05)
    wq = pk.fragment_major("wf")
    ref = K.conv_forward(x, pk.wf, (3, 3, 3), 64).clone()
    n_bad = sum(0 if torch.equal(ref, K.conv_forward(x, pk.wf, (3, 3, 3), 64, wq=wq)) else 1 for _ in range(reps))
    print(f"3-D 16x12x32x32 64->64: {n_bad}/{reps} runs differ")
    print("TOTAL differing runs:", bad + n_bad)


if __name__ == "__main__":
    main()
