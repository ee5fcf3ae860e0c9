"""Size-independent properties of the convolution kernels at the FULL shapes of BASELINE configs[1]
(B=64, T=48: 3072 frames per batched launch), where a CPU reference would take minutes:

  adjoint identity   <conv(x; w), y> = <x, conv^T(y; w)> = <w, wgrad(x, y)>   ties forward, backward-data and
                     backward-weight together (all three are separate kernels / weight packs);
  batch locality     the rows of a frame do not depend on which other frames are in the launch (bitwise);
  exact scaling      conv(2x) = 2 conv(x) bitwise (a power of two commutes with every rounding);
  additivity         wgrad over the batch = sum of wgrad over its two halves.

Shapes: the batched ConvGRU x-path conv at 32x32 (halo kernel, filter-row weight gradient, 5x5), the last
GResBlock conv at 64x64 (thin 64-channel tile, 3x3), and the recurrent 16x16 conv that runs split-K.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [   # frames, S, Cin, Cout, k
    (3072, 32, 256, 384, 5),
    (3072, 64, 64, 64, 3),
    (64, 16, 1024, 512, 5),
]


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("shape", SHAPES)
def test_adjoint_locality_scaling_additivity(shape):
    from dvd_gan_amd import kern as K
    F_, S, Cin, Cout, k = shape
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(F_ + S + Cin)
    x = torch.randn(F_, S, S, Cin, device=dev, generator=g).to(dt)
    y = torch.randn(F_, S, S, Cout, device=dev, generator=g).to(dt)
    w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5
    pk = K.PackedConv(dt, Cout, Cin, (k, k), dev).fill(w)
    wq = pk.wf.float().permute(1, 2, 0).reshape(Cout, Cin, k, k)          # the bf16-rounded weights the kernels see
    out = K.conv_forward(x, pk.wf, (k, k), Cout, out_f32=True)            # [F,S,S,Cout] fp32
    dx = K.conv_forward(y, pk.wd, (k, k), pk.cip, out_f32=True)           # conv^T through the flipped pack
    dw = torch.zeros(Cout, Cin, k, k, device=dev)
    K.conv_wgrad(x, y, dw, (k, k), Cout, Cin)
    a, b, c = dot(out, y), dot(x, dx[..., :Cin]), dot(wq, dw)
    scale = float(out.double().norm() * y.double().norm())
    assert abs(a - b) < 2e-6 * scale and abs(a - c) < 2e-6 * scale, (a, b, c, scale)

    # batch locality: frames [f0, f0 + n) computed alone are bitwise the same rows
    f0, n = F_ // 3, max(1, F_ // 8)
    sub = K.conv_forward(x[f0:f0 + n].contiguous(), pk.wf, (k, k), Cout, out_f32=True)
    assert torch.equal(sub, out[f0:f0 + n])

    # exact scaling by a power of two
    out2 = K.conv_forward((x.float() * 2).to(dt), pk.wf, (k, k), Cout, out_f32=True)
    assert torch.equal(out2, out * 2)

    # additivity of the weight gradient over the batch (different row-split plans -> fp32 rounding only)
    h = F_ // 2
    dw1, dw2 = torch.zeros_like(dw), torch.zeros_like(dw)
    K.conv_wgrad(x[:h].contiguous(), y[:h].contiguous(), dw1, (k, k), Cout, Cin)
    K.conv_wgrad(x[h:].contiguous(), y[h:].contiguous(), dw2, (k, k), Cout, Cin)
    assert float((dw1 + dw2 - dw).double().norm() / dw.double().norm()) < 1e-5


def test_full_size_step_invariants():
    """One G + D_s + D_t step at the benchmark configuration (B=64, T=48, 64x64, ch=32, bf16): finite losses in the
    hinge range, generator output in (-1, 1) with the reference shape, every parameter moved by at most lr-sized
    Adam steps, spectral-norm vectors still unit length."""
    import argparse
    from dvd_gan_amd.train_step import Trainer
    torch.manual_seed(0)
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=64, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    tr = Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.bfloat16)
    before = [p.detach().clone() for p in tr.G.parameters() if p.requires_grad]
    real = torch.rand(64, 3, 48, 64, 64) * 2 - 1
    labels = torch.randint(0, 101, (64,))
    losses = [float(v.detach()) for v in tr.train_step(real, labels)]
    assert all(l == l and 0.0 <= l < 10.0 for l in losses), losses        # hinge terms are >= 0
    with torch.no_grad():
        fake = tr.G(torch.randn(4, 120, device="cuda"), torch.randint(0, 101, (4,), device="cuda"))
    assert fake.shape == (4, 48, 3, 64, 64) and float(fake.abs().max()) <= 1.0
    after = [p.detach() for p in tr.G.parameters() if p.requires_grad]
    step = max(float((a - b).abs().max()) for a, b in zip(after, before))
    assert 0.0 < step <= 5e-5 * 1.01 * 3.2, step       # Adam, beta1 = 0, first step: |delta| = lr * |g| / (|g| sqrt(0.1) + eps) <= lr / sqrt(0.1)
    for name, buf in tr.D_s.state_dict().items():
        if name.endswith(("weight_u", "weight_v")):
            assert abs(float(buf.norm()) - 1.0) < 1e-3, name
