"""Times single conv_igemm / conv_wgrad launches on representative C2 shapes (HIP events).
usage: python tools/conv_microbench.py [fwd|wgrad|all] [iters] [shape indices, e.g. 3,4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import kern as K

SHAPES = [   # frames, S, Cin, Cout, k, split-K slabs?
    (64, 32, 512, 256, 5, True),      # GRU#3 h-path, serial step
    (64, 16, 1024, 512, 5, True),     # GRU#2 h-path
    (64, 8, 512, 512, 5, True),       # GRU#1 h-path (split-K)
    (3072, 32, 256, 384, 5, False),   # GRU#3 batched x-path
    (3072, 16, 256, 1536, 5, False),  # GRU#2 batched x-path (wide N)
    (3072, 16, 1536, 256, 5, False),  # its backward-data
    (3072, 16, 256, 256, 3, False),   # GResBlock 3x3
    (3072, 32, 128, 128, 3, False),
    (64, 32, 128, 256, 3, False),     # 8: gru3.l0 [u|r] h-path, one step
    (64, 32, 128, 128, 3, False),     # 9: gru3.l0 out-gate h-path
    (64, 16, 256, 512, 3, False),     # 10: gru2.l0 [u|r]
    (64, 16, 256, 256, 3, False),     # 11
    (64, 32, 256, 512, 5, False),     # 12: gru3.l1 [u|r]
    (64, 8, 256, 512, 3, False),      # 13: gru1.l0 [u|r]
    (64, 4, 256, 512, 3, False),      # 14: gru0.l0 [u|r]
    (64, 32, 32, 256, 3, False),      # 15..18: K sweep at fixed M / Cout (fixed cost vs per-K cost of a one-round launch)
    (64, 32, 64, 256, 3, False),
    (64, 32, 256, 256, 3, False),
    (64, 32, 512, 256, 3, False),
    (128, 32, 128, 256, 3, False),    # 19, 20: two and four rounds of the same tiles
    (256, 32, 128, 256, 3, False),
    (3008, 32, 256, 256, 5, False),   # 21..24: the large weight-gradient shapes (gru3.l1 / gru2.l1 h-part, 3x3 blocks)
    (3008, 16, 512, 512, 5, False),
    (3072, 32, 128, 128, 3, False),
    (3072, 16, 256, 256, 3, False),
    (3072, 32, 128, 256, 5, False),   # 25..27: second tier of the weight-gradient shapes
    (3072, 16, 256, 512, 5, False),
    (3072, 32, 256, 128, 5, False),
    (3072, 32, 128, 768, 5, False),   # 28: gru3.l1 x-part (short K, wide N)
    (3072, 32, 768, 128, 5, False),   # 29: its backward-data
    (3072, 64, 64, 64, 3, False),     # 30: thin output (256 x 64 tile): last GResBlock conv1
    (3072, 64, 128, 64, 3, False),    # 31
    (512, 64, 64, 64, 3, False),      # 32: D_s pre_conv.2
    (3008, 4, 512, 512, 5, False),    # 33..36: weight gradients on 4 x 4 frames (gru0.l1 h-part / x-part, gru0.l0, GResBlock)
    (3072, 4, 256, 1536, 5, False),
    (3008, 4, 256, 256, 3, False),
    (3072, 4, 256, 256, 3, False),
    (3072, 64, 128, 64, 3, False, True),   # 37..39: 3 x 3 over a nearest-x2 upsampled input (GResBlock conv1), S = the OUTPUT size
    (3072, 32, 256, 128, 3, False, True),
    (3072, 16, 256, 256, 3, False, True),
]


def bench(fn, iters):
    """`iters` back-to-back launches replayed from a HIP graph (no Python / ctypes time between launches: the recurrent
    convolutions take 15-70 us, less than a ctypes call)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev, dt = "cuda", torch.bfloat16
    sel = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(len(SHAPES))
    for shp in [SHAPES[i] for i in sel]:
        F_, S, Cin, Cout, k, slabs = shp[:6]
        up2 = len(shp) > 6 and shp[6]
        if up2:
            if what in ("wgrad", "all"):
                x = torch.randn(F_, S // 2, S // 2, Cin, device=dev).to(dt)
                dy = torch.randn(F_, S, S, Cout, device=dev).to(dt)
                dw = torch.zeros(Cout, Cin, k, k, device=dev)
                db = torch.zeros(Cout, device=dev) if os.environ.get("WG_BIAS") == "1" else None
                M, fl = F_ * S * S, 2.0 * F_ * S * S * Cout * Cin * k * k
                ms = bench(lambda: K.conv_wgrad(x, dy, dw, (k, k), Cout, Cin, dbias=db, up2=True), max(2, iters // 3))
                print(f"wgrad M={M:8d} C={Cin:5d} Cout={Cout:5d} k={k} up2      : {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
            continue
        x = torch.randn(F_, S, S, Cin, device=dev).to(dt)
        pk = K.PackedConv(dt, Cout, Cin, (k, k), dev).fill(torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
        M = F_ * S * S
        fl = 2.0 * M * Cout * Cin * k * k
        if what in ("fwd", "all"):
            import ctypes as C
            from dvd_gan_amd import lib as L
            ns = L.lib().dvd_conv_pick_nsplit(L.BF16, C.c_longlong(M), Cout, Cin, k * k) if slabs else 1
            ws = torch.empty(ns, M, Cout, device=dev) if slabs else None
            out = None if slabs else torch.empty(F_, S, S, Cout, device=dev, dtype=dt)
            if os.environ.get("GB") == "1":        # interleaved A/B: weights through LDS vs straight from L2 (fragment-major image)
                wq = pk.fragment_major("wf")
                ref = K.conv_forward(x, pk.wf, (k, k), Cout, nsplit=ns, slabs=slabs)
                got = K.conv_forward(x, pk.wf, (k, k), Cout, nsplit=ns, slabs=slabs, wq=wq)
                same = bool(torch.equal(ref, got))
                res = []
                for _ in range(3):
                    a_ = bench(lambda: K.conv_forward(x, pk.wf, (k, k), Cout, nsplit=ns, ws=ws, out=out), iters)
                    b_ = bench(lambda: K.conv_forward(x, pk.wf, (k, k), Cout, nsplit=ns, ws=ws, out=out, wq=wq), iters)
                    res.append((a_, b_))
                a_, b_ = min(r[0] for r in res), min(r[1] for r in res)
                print(f"fwd   M={M:8d} C={Cin:5d} Cout={Cout:5d} k={k} split={ns:2d}: lds {a_ * 1e3:9.1f} us {fl / a_ / 1e9:7.1f} TF/s | "
                      f"gb {b_ * 1e3:9.1f} us {fl / b_ / 1e9:7.1f} TF/s  ({(a_ / b_ - 1) * 100:+.1f} %)  identical={same}", flush=True)
                continue
            ms = bench(lambda: K.conv_forward(x, pk.wf, (k, k), Cout, nsplit=ns, ws=ws, out=out), iters)
            print(f"fwd   M={M:8d} C={Cin:5d} Cout={Cout:5d} k={k} split={ns:2d}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
        if what in ("wgrad", "all") and not slabs:
            dy = torch.randn(F_, S, S, Cout, device=dev).to(dt)
            dw = torch.zeros(Cout, Cin, k, k, device=dev)
            db = torch.zeros(Cout, device=dev) if os.environ.get("WG_BIAS") == "1" else None      # bias gradient rides along
            ms = bench(lambda: K.conv_wgrad(x, dy, dw, (k, k), Cout, Cin, dbias=db), max(2, iters // 3))
            print(f"wgrad M={M:8d} C={Cin:5d} Cout={Cout:5d} k={k}          : {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
