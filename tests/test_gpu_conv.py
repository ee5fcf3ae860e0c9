"""GPU parity of the implicit-GEMM convolution kernels (forward, backward-data via the flipped
pack, backward-weight) through the C ABI against a plain fp32 torch reference of the same op.

Tolerances (rel-L2 = |a-b|_2 / |b|_2):
  exact mode (f32 MFMA)                      <= 2e-6
  bf16 mode vs reference on bf16-ROUNDED inputs, fp32 output   <= 2e-6  (only summation order differs)
  bf16 mode vs the unrounded fp32 reference  <= 4e-3  
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def bf(x):
    return x.to(torch.bfloat16).float()


def ref_conv(x, w, b=None, up2=False, relu_in=False):
    if relu_in:
        x = F.relu(x)
    if up2:
        x = F.interpolate(x, scale_factor=2)
    pad = [k // 2 for k in w.shape[2:]]
    return (F.conv3d if w.dim() == 5 else F.conv2d)(x, w, b, padding=pad)


CASES = [
    # frames, Cin, Cout, spatial, ksize, up2, relu_in
    (3, 16, 24, (8, 8), (3, 3), False, False),
    (5, 40, 136, (4, 4), (5, 5), False, False),
    (2, 8, 8, (16, 16), (1, 1), False, True),
    (4, 24, 40, (8, 8), (3, 3), True, True),
    (2, 3, 16, (32, 32), (3, 3), False, False),      # 3 input channels padded to 8
    (2, 16, 3, (16, 16), (3, 3), False, True),       # 3 output channels
    (2, 8, 16, (6, 8, 8), (3, 3, 3), False, False),  # 3-D, T=6
    (1, 16, 8, (4, 4, 4), (1, 1, 1), False, False),
    (7, 136, 264, (4, 4), (3, 3), False, False),     # several K chunks, 3 N tiles, ragged M
    (64, 16, 256, (32, 32), (3, 3), False, False),   # >= 256 tiles of 256x256: the 8-wave kernel
    (66, 8, 512, (32, 32), (1, 1), False, True),     # 8-wave kernel, ReLU variant, ragged M, 2 N tiles
    # halo-staged kernel (frames >= one 16-wide patch, 3x3 / 5x5 filters)
    (3, 72, 40, (16, 16), (5, 5), False, False),     # 128-pixel patches, three K chunks (split-K over chunks)
    (2, 24, 40, (16, 16), (3, 3), True, True),       # nearest x2 folded into the footprint, ReLU on fragments
    (2, 16, 24, (16, 64), (3, 3), False, False),     # H != W
    (1, 8, 16, (4, 16, 16), (3, 3, 3), False, False),  # 3-D: footprint from frame t+dt
    (32, 16, 40, (64, 64), (5, 5), False, True),     # 512 tiles, Cout <= 64: the 256 x 64 thin-output variant
    (8, 24, 40, (64, 64), (3, 3), True, False),      # thin-output variant with the x2 upsample folded in
    (32, 16, 72, (64, 64), (3, 3), False, True),     # 512 tiles of 256 x 128: the 16 x 16 patch variant
    (16, 40, 264, (64, 64), (5, 5), False, False),   # 8-wave 256 x 256 variant, 5x5, ragged N
    # filter-row weight-gradient kernel (Cout >= 96, Cin >= 48, W >= 16)
    (3, 72, 136, (16, 16), (5, 5), False, False),    # W = 16: a 32-pixel step is two lines; 128-channel tile, ragged
    (2, 64, 200, (32, 32), (3, 3), False, True),     # 256-channel tile, 3 taps per row, ReLU on the staged input
    (2, 56, 264, (16, 64), (5, 5), False, False),    # W = 64: two segments per line; 2 x 1 tiles, ragged both ways
    # more filter-row geometries: each tile width x filter size x line geometry, staggered 8-wave variants
    (5, 48, 200, (8, 8), (5, 5), False, False),      # W = 8: four lines per step, 256-channel tile
    (9, 72, 56, (8, 16), (3, 3), False, False),      # 64-channel tile, ragged slices
    (1, 16, 136, (4, 16, 16), (3, 3, 3), False, False),  # 3-D, 256-channel tile: footprint from frame t+dt
    (33, 24, 120, (32, 32), (5, 5), False, False),   # 128-channel tile, 64 input channels, several ragged row slices
    (6, 128, 120, (16, 16), (5, 5), False, False),   # 128 x 128-channel tile (8 waves), 5 taps
    (2, 256, 384, (32, 32), (3, 3), False, True),    # 128 x 128-channel tile, 3 x 2 tiles, ReLU on the staged input
    # one-wave-per-SIMD filter-row kernel (round 6: 3 taps, Cout > 64, input channels in 128-wide tiles): its line geometries
    (5, 128, 136, (8, 8), (3, 3), False, True),      # W = 8: four lines per step, ReLU on the staged input, 2 x 1 tiles
    (2, 72, 136, (16, 64), (3, 3), False, False),    # W = 64: two 32-pixel segments per line, H != W, ragged input tile
    (3, 192, 264, (32, 32), (3, 3), False, False),   # W = 32, 3 x 2 tiles, several row slices
    (1, 128, 128, (3, 16, 16), (3, 3, 3), False, False),   # 3-D, T = 3
    # x2-upsampled input folded onto the input grid (round 6: conv_wgrad_row4_kernel<.., FOLD>, taken with a forced split here):
    # the line geometries of the INPUT grid, phase blocks of 64 / 128 / 192 channels inside the 128-channel tiles
    (4, 128, 64, (64, 64), (3, 3), True, True),      # input lines of 32 pixels: one line per step; a tile = two column phases
    (2, 72, 128, (32, 32), (3, 3), True, False),     # 16-pixel input lines: two per step; a tile = one phase; ragged input tile
    (3, 128, 192, (16, 16), (3, 3), True, True),     # 8-pixel input lines: four per step; tiles straddle the column phases
    (6, 200, 64, (8, 8), (3, 3), True, False),       # 4 x 4 input frames: a step = two frames; two input-channel tiles
    (2, 128, 64, (16, 128), (3, 3), True, True),     # 64-pixel input lines: two segments per line, H != W
    # nine-tap 3 x 3 weight-gradient kernel (Cout > 64): the three line geometries of its footprint
    (5, 40, 136, (8, 8), (3, 3), False, False),      # W = 8: four lines per step + 2 halo lines of 10 rows
    (3, 72, 200, (16, 16), (3, 3), False, True),     # W = 16: two lines + 2, two in-channel tiles, ReLU
    (2, 24, 72, (16, 64), (3, 3), False, False),     # W = 64: two 32-pixel segments per line, H != W
    (1, 72, 136, (6, 8, 16), (3, 3, 3), False, False),   # 3-D, T = 6
    # pixel-major row order of the tap-by-tap kernel (frames a multiple / divisor of the tile height, 4 x 4 and 8 x 8 frames):
    # filter rows outside the frame are skipped per tile
    (64, 40, 136, (4, 4), (5, 5), False, False),         # 128-row tiles = 2 pixels x 64 frames; ragged N, split-K over fewer steps
    (128, 16, 24, (8, 8), (3, 3), False, True),          # one pixel per tile
    (32, 24, 264, (8, 8), (5, 5), False, False),         # 4 pixels x 32 frames, three N tiles
    (256, 16, 264, (4, 4), (5, 5), False, False),        # 8-wave 256 x 256 tile
    # filter-row weight-gradient kernel on 4 x 4 frames (round 4): a 32-pixel step = two whole frames
    (6, 72, 200, (4, 4), (3, 3), False, True),
    (10, 136, 72, (4, 4), (5, 5), False, False),
    # the thin ends of the networks in bf16 mode (wgrad_thin.hip: 3 channels on one side, 64 on the other, taps folded into the
    # matrix dimension; exact mode takes the general kernels on the same cases)
    (5, 3, 64, (32, 32), (3, 3), False, False),          # a discriminator stem: x thin, bias gradient through the 1.0 column
    (2, 3, 64, (5, 16, 16), (3, 3, 3), False, True),     # 3-D stem, T = 5, ReLU on the thin input
    (3, 64, 3, (64, 64), (3, 3), False, True),           # the RGB layer: dy thin (taps mirrored), ReLU on the wide input
    (2, 64, 2, (3, 32, 32), (3, 3, 3), False, False),    # dy thin, 3-D
    # extents that are not powers of two (latent_dim 3 / 6: 6, 12, 24, 48, 96 pixels): division indexing, tap-by-tap kernels
    (3, 16, 24, (6, 6), (3, 3), False, False),
    (2, 24, 40, (12, 12), (5, 5), False, True),
    (2, 16, 40, (24, 12), (3, 3), True, True),           # x2 upsample folded in, H != W
    (5, 40, 264, (6, 6), (5, 5), False, False),          # wide N, ragged M
    (1, 8, 16, (3, 6, 6), (3, 3, 3), False, False),      # 3-D
    (2, 24, 72, (48, 48), (3, 3), False, False),         # a multiple of 16 that is not a power of two
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_forward_dgrad_wgrad(case, dtype):
    from dvd_gan_amd import kern as K
    F_, Cin, Cout, sp, ks, up2, relu_in = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(F_, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, *ks, generator=g) / (Cin * ks[-1] * ks[-2]) ** 0.5
    b = torch.randn(Cout, generator=g)
    dev = "cuda"
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = ref_conv(xr, wr, b, up2, relu_in)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    # rounded-input reference for the tight bf16 check
    exact = dtype == torch.float32
    xq, wq, gyq = (x, w, gy) if exact else (bf(x), bf(w), bf(gy))
    xq_ = xq.clone().requires_grad_(True)
    wq_ = wq.clone().requires_grad_(True)
    yq = ref_conv(xq_, wq_, b, up2, relu_in)
    yq.backward(gyq)

    xc = K.to_cl(x.to(dev), dtype)
    pk = K.PackedConv(dtype, Cout, Cin, ks, dev).fill(w.to(dev))
    # forward, fp32 output for the tight check
    y = K.conv_forward(xc, pk.wf, ks, Cout, bias=b.to(dev), up2=up2, relu_in=relu_in, out_f32=True)
    y_nc = K.from_cl(y, Cout).cpu()
    assert rel(y_nc, yq.detach()) < 2e-6
    assert rel(y_nc, y_ref.detach()) < (2e-6 if exact else 4e-3)
    # split-K slabs sum to the same result (bias excluded)
    nsplit = 3
    ws = K.conv_forward(xc, pk.wf, ks, Cout, up2=up2, relu_in=relu_in, nsplit=nsplit, slabs=True)
    y_split = ws.sum(0).view(*y.shape[:-1], Cout) + b.to(dev)
    assert rel(y_split.cpu(), y_nc.movedim(1, -1)) < 2e-6
    # storage-dtype output + tanh + residual + mask epilogue
    res = torch.randn(y_ref.shape, generator=g)
    msk = torch.randn(y_ref.shape, generator=g)
    resc, mskc = K.to_cl(res.to(dev), dtype), K.to_cl(msk.to(dev), dtype)
    y2 = K.conv_forward(xc, pk.wf, ks, Cout, bias=b.to(dev), res=resc, mask=mskc, act=2, up2=up2, relu_in=relu_in)
    resq, mskq = (res, msk) if exact else (bf(res), bf(msk))
    want = torch.tanh(yq.detach() + resq) * (mskq > 0)
    assert rel(K.from_cl(y2, Cout).cpu(), want) < (2e-6 if exact else 3e-3)
    if Cout % 8:
        assert float(y2[..., Cout:].abs().max()) == 0.0     # pad channels stay zero

    # backward-data = forward kernel on the flipped / transposed pack
    gyc = K.to_cl(gy.to(dev), dtype)
    dx = K.conv_forward(gyc, pk.wd, ks, pk.cip, out_f32=True)
    dx_nc = K.from_cl(dx, Cin).cpu()
    if up2:    # gradient of nearest x2 upsample = 2x2 sum
        dx_nc = F.avg_pool2d(dx_nc, 2) * 4
    if relu_in:
        dx_nc = dx_nc * (xq > 0)
    assert rel(dx_nc, xq_.grad) < 3e-6
    assert rel(dx_nc, xr.grad) < (3e-6 if exact else 6e-3)
    if up2:    # the same 2 x 2 sums (+ ReLU mask) formed in the conv's epilogue (round 6: dvd_conv_desc.pool2) where a halo kernel serves it
        fused = K.conv_forward(gyc, pk.wd, ks, pk.cip, out_f32=True, pool2=True, mask=xc if relu_in else None,
                               wq=lambda: pk.fragment_major("wd"))
        assert (fused is not None) == (len(sp) == 2 and min(sp) >= 8 and all(v & (v - 1) == 0 for v in sp)), sp
        if fused is not None:
            assert fused.shape[1:3] == (sp[0], sp[1])
            assert rel(K.from_cl(fused, Cin).cpu(), xq_.grad) < 3e-6
            fused_b = K.conv_forward(gyc, pk.wd, ks, pk.cip, pool2=True, mask=xc if relu_in else None)      # storage dtype, weights through LDS
            assert rel(K.from_cl(fused_b, Cin).cpu().float(), xq_.grad) < (3e-6 if exact else 4e-3)

    # backward-weight (fp32 atomics into the reference layout)
    dw = torch.zeros(Cout, Cin, *ks, device=dev)
    K.conv_wgrad(xc, gyc, dw, ks, Cout, Cin, up2=up2, relu_in=relu_in)
    assert rel(dw.cpu(), wq_.grad) < 5e-6
    assert rel(dw.cpu(), wr.grad) < (5e-6 if exact else 6e-3)
    # fused bias gradient (column sums of dy, accumulated by the centre-row workgroups)
    db = torch.zeros(Cout, device=dev)
    K.conv_wgrad(xc, gyc, torch.zeros_like(dw), ks, Cout, Cin, up2=up2, relu_in=relu_in, dbias=db)
    assert rel(db.cpu(), gyq.transpose(0, 1).reshape(Cout, -1).sum(1)) < 5e-6
    dw2 = torch.zeros_like(dw)
    K.conv_wgrad(xc, gyc, dw2, ks, Cout, Cin, up2=up2, relu_in=relu_in, msplit=1)
    # one slice = one fp32 accumulator chain over all M rows: rounding grows with the chain length (131072 rows: 6.5e-6)
    assert rel(dw2.cpu(), wq_.grad) < 1e-5
    # a caller-forced split also takes the one-wave-per-SIMD tiles (round 6) on shapes the planner would leave to the 8-wave tiles
    # for lack of rows: two slices through the workspace, with and without the bias column sums
    if int(torch.tensor(sp).prod()) * F_ >= 64:
        dw3, db3 = torch.zeros_like(dw), torch.zeros(Cout, device=dev)
        K.conv_wgrad(xc, gyc, dw3, ks, Cout, Cin, up2=up2, relu_in=relu_in, msplit=2, dbias=db3)
        assert rel(dw3.cpu(), wq_.grad) < 1e-5
        assert rel(db3.cpu(), gyq.transpose(0, 1).reshape(Cout, -1).sum(1)) < 5e-6
        dw4 = torch.zeros_like(dw)
        K.conv_wgrad(xc, gyc, dw4, ks, Cout, Cin, up2=up2, relu_in=relu_in, msplit=2)
        assert rel(dw4.cpu(), wq_.grad) < 1e-5


@pytest.mark.parametrize("case", [(5, (64, 64), (3, 3), True, 1, False), (3, (6, 32, 32), (3, 3, 3), False, 0, False),
                                  (4, (32, 32), (3, 3), False, 1, True), (2, (4, 64, 64), (3, 3, 3), True, 0, True)])
def test_thin_input_convolution(case):
    """3 -> 64 channels in bf16 mode (the discriminator stems; with a mask: the backward-data pass of the RGB layer): the kernel
    that folds the KW taps into the K dimension (conv_thin.hip), against torch on the bf16-rounded operands and against the
    halo-staged kernel (no weight image supplied)."""
    from dvd_gan_amd import kern as K
    from dvd_gan_amd import lib as L
    F_, sp, ks, relu_in, act, use_mask = case
    g = torch.Generator().manual_seed(41)
    x = torch.randn(F_, 3, *sp, generator=g)
    w = torch.randn(64, 3, *ks, generator=g) / (3 * ks[-1] * ks[-2]) ** 0.5
    b = torch.randn(64, generator=g)
    msk = torch.randn(F_, 64, *sp, generator=g)
    dev = "cuda"
    want = ref_conv(bf(x), bf(w), b, False, relu_in)
    if act:
        want = F.relu(want)
    if use_mask:
        want = want * (bf(msk) > 0)
    xc = K.to_cl(x.to(dev), torch.bfloat16)
    mc = K.to_cl(msk.to(dev), torch.bfloat16) if use_mask else None
    pk = K.PackedConv(torch.bfloat16, 64, 3, ks, dev).fill(w.to(dev))
    import ctypes as C
    got = K.conv_forward(xc, pk.wf, ks, 64, bias=b.to(dev), act=act, relu_in=relu_in, mask=mc, wq=lambda: pk.fragment_major("wf"))
    assert getattr(pk.wf, "_thin_img", None) is not None, "the request should have taken the thin-input kernel"
    ref = K.conv_forward(xc, pk.wf, ks, 64, bias=b.to(dev), act=act, relu_in=relu_in, mask=mc)
    assert rel(K.from_cl(got, 64).cpu(), want) < 3e-3
    assert rel(got.float().cpu(), ref.float().cpu()) < 1e-3       # same operands, fp32 sums in another order, one bf16 rounding
    # the sums BEFORE the bf16 rounding (out_f32, as every other convolution is held to): only fp32 summation order is left
    got32 = K.conv_forward(xc, pk.wf, ks, 64, bias=b.to(dev), act=act, relu_in=relu_in, mask=mc, out_f32=True,
                           wq=lambda: pk.fragment_major("wf"))
    assert got32.dtype == torch.float32
    assert rel(K.from_cl(got32, 64).cpu(), want) < 2e-6


@pytest.mark.parametrize("case", [(5, 64, True, 2), (3, 32, False, 0), (2, 64, False, 1)])
def test_thin_output_convolution(case):
    """64 -> 3 channels, 3 x 3, bf16 mode (the generator's RGB layer: ReLU -> conv -> tanh) and, on the backward-data pack of a
    3 -> 64 stem with a ReLU mask, 64 -> 8 (3 real): conv_thin.hip's thin-output kernel against torch on the bf16-rounded operands
    and against the halo-staged kernel."""
    from dvd_gan_amd import kern as K
    F_, S, relu_in, act = case
    g = torch.Generator().manual_seed(43)
    dev = "cuda"
    x = torch.randn(F_, 64, S, S, generator=g)
    w = torch.randn(3, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(3, generator=g)
    want = ref_conv(bf(x), bf(w), b, False, relu_in)
    want = torch.tanh(want) if act == 2 else F.relu(want) if act == 1 else want
    xc = K.to_cl(x.to(dev), torch.bfloat16)
    pk = K.PackedConv(torch.bfloat16, 3, 64, (3, 3), dev).fill(w.to(dev))
    got = K.conv_forward(xc, pk.wf, (3, 3), 3, bias=b.to(dev), act=act, relu_in=relu_in, wq=lambda: pk.fragment_major("wf"))
    assert getattr(pk.wf, "_thin_img", None) is not None, "the request should have taken the thin-output kernel"
    ref = K.conv_forward(xc, pk.wf, (3, 3), 3, bias=b.to(dev), act=act, relu_in=relu_in)
    assert rel(K.from_cl(got, 3).cpu(), want) < 4e-3
    assert rel(got.float().cpu(), ref.float().cpu()) < 2e-3
    assert float(got[..., 3:].abs().max()) == 0.0                   # pad channels stay zero
    got32 = K.conv_forward(xc, pk.wf, (3, 3), 3, bias=b.to(dev), act=act, relu_in=relu_in, out_f32=True, wq=lambda: pk.fragment_major("wf"))
    assert got32.dtype == torch.float32 and float(got32[..., 3:].abs().max()) == 0.0
    assert rel(K.from_cl(got32, 3).cpu(), want) < (2e-6 if act != 2 else 2e-5)     # (tanh: the bf16 mode's hardware exp / rcp form, ~1e-6 relative per value)
    # backward-data of a 3 -> 64 stem: dy [.., 64] through the flipped / transposed pack, masked by the stem's input
    ws = torch.randn(64, 3, 3, 3, generator=g) / 5.0
    xin = torch.randn(F_, 3, S, S, generator=g)
    ps = K.PackedConv(torch.bfloat16, 64, 3, (3, 3), dev).fill(ws.to(dev))
    mc = K.to_cl(xin.to(dev), torch.bfloat16)
    dx = K.conv_forward(xc, ps.wd, (3, 3), ps.cip, mask=mc, wq=lambda: ps.fragment_major("wd"))
    assert getattr(ps.wd, "_thin_img", None) is not None
    dx_ref = K.conv_forward(xc, ps.wd, (3, 3), ps.cip, mask=mc)
    wdx = torch.nn.functional.conv_transpose2d(bf(x), bf(ws), padding=1) * (bf(xin) > 0)
    assert rel(K.from_cl(dx, 3).cpu(), wdx) < 4e-3
    assert rel(dx.float().cpu(), dx_ref.float().cpu()) < 2e-3
    dx32 = K.conv_forward(xc, ps.wd, (3, 3), ps.cip, mask=mc, out_f32=True, wq=lambda: ps.fragment_major("wd"))
    assert rel(K.from_cl(dx32, 3).cpu(), wdx) < 2e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 16, 40, (16, 16), (3, 3)), (2, 8, 24, (4, 8, 8), (3, 3, 3)), (5, 24, 8, (32, 32), (1, 1)),
                                   (3, 16, 24, (12, 12), (3, 3)), (2, 8, 8, (6, 24), (1, 1))])
def test_residual_read_through_nearest_upsample(shape, dtype):
    """res_up2: the residual operand lives on the H/2 x W/2 grid and is expanded (nearest x2) in the epilogue --
    out = conv(x) + b + upsample(res)."""
    from dvd_gan_amd import kern as K
    F_, Cin, Cout, sp, ks = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(F_, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, *ks, generator=g) / (Cin * ks[-1] * ks[-2]) ** 0.5
    b = torch.randn(Cout, generator=g)
    half = sp[:-2] + (sp[-2] // 2, sp[-1] // 2)
    res = torch.randn(F_, Cout, *half, generator=g)
    exact = dtype == torch.float32
    xq, wq, rq = (x, w, res) if exact else (bf(x), bf(w), bf(res))
    up = F.interpolate(rq, scale_factor=(1, 2, 2) if len(sp) == 3 else 2)
    want = ref_conv(xq, wq, b) + up
    dev = "cuda"
    pk = K.PackedConv(dtype, Cout, Cin, ks, dev).fill(w.to(dev))
    y = K.conv_forward(K.to_cl(x.to(dev), dtype), pk.wf, ks, Cout, bias=b.to(dev), res=K.to_cl(res.to(dev), dtype),
                       res_up2=True, out_f32=True)
    assert rel(K.from_cl(y, Cout).cpu(), want) < 2e-6


def test_bad_shapes_raise():
    from dvd_gan_amd import kern as K
    x = torch.zeros(1, 5, 8, 8, device="cuda")          # an odd extent cannot be the output of a x2 upsample
    pk = K.PackedConv(torch.float32, 8, 8, (3, 3), "cuda")
    with pytest.raises(RuntimeError):
        K.conv_forward(torch.zeros(1, 5, 4, 8, device="cuda"), pk.wf, (3, 3), 8, res=x, res_up2=True)
    pk2 = K.PackedConv(torch.float32, 8, 8, (2, 2), "cuda")     # even filter
    with pytest.raises(RuntimeError):
        K.conv_forward(torch.zeros(1, 8, 8, 8, device="cuda"), pk2.wf, (2, 2), 8)


def test_f32_to_bf16_is_round_to_nearest_even():
    """The kernels round with the hardware conversion (v_cvt_pk_bf16_f32): bit-equal to torch's RNE cast on random bit
    patterns, ties, denormals, the overflow boundary and infinities; NaN stays NaN."""
    from dvd_gan_amd import kern as K
    g = torch.Generator().manual_seed(3)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 << 16,), generator=g, dtype=torch.int64).to(torch.int32)
    special = torch.tensor([0x00000000, 0x80000000, 0x3f808000, 0x3f818000, 0x3f807fff, 0x3f808001, 0x00000001, 0x00008000,
                            0x00018000, 0x007fffff, 0x7f7f7fff, 0x7f7f8000, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000,
                            0x7f800001], dtype=torch.int64).to(torch.int32)
    x = torch.cat([bits, special]).view(torch.float32)
    x = torch.cat([x, torch.zeros((-x.numel()) % 8)])
    got = K.convert(x.cuda(), torch.bfloat16).cpu()
    want = x.to(torch.bfloat16)
    nan = torch.isnan(want.float())
    assert bool((torch.isnan(got.float()) == nan).all())
    assert torch.equal(got.view(torch.int16)[~nan], want.view(torch.int16)[~nan])


DEV = "cuda"


GB_CASES = [  # frames, S, Cin, Cout, k, up2, relu_in, nsplit
    (64, 32, 512, 256, 5, 0, 0, 1), (64, 16, 1024, 512, 5, 0, 0, 2), (256, 32, 256, 384, 5, 0, 0, 1),      # 256 x 128 tile
    (64, 32, 128, 128, 3, 0, 0, 1), (64, 16, 256, 256, 3, 0, 0, 2),                                          # 128 x 128 tile
    (512, 64, 64, 64, 3, 0, 0, 1), (512, 64, 8, 64, 3, 0, 0, 1), (512, 64, 128, 64, 3, 1, 1, 1),             # 256 x 64 (thin) tile
    (256, 32, 256, 128, 3, 1, 1, 1), (256, 32, 128, 128, 3, 0, 1, 1),                                        # x2 fold, ReLU on fragments
    (64, 8, 512, 1024, 5, 0, 0, 4), (64, 8, 256, 512, 3, 0, 0, 4), (3072, 8, 256, 256, 3, 0, 1, 1),         # whole-frame footprints, 8 x 8
    (64, 4, 512, 1024, 5, 0, 0, 8), (3072, 4, 256, 768, 3, 0, 0, 1), (61, 8, 64, 96, 3, 0, 0, 1), (13, 4, 40, 72, 5, 0, 1, 2),
]


@pytest.mark.parametrize("case", GB_CASES)
def test_weights_from_l2_kernels_equal_the_lds_staged_kernels(case):
    """conv_halo_gb_kernel / conv_halo_gbs_kernel (weight fragments read from L2 in fragment-major order, no per-tap barrier) keep
    the K order of conv_halo_kernel / conv_igemm_kernel: the outputs must be BIT-identical, on every tile shape, with the nearest
    x2 fold, ReLU on the fragments, split-K slabs, ragged frame counts and channel counts -- repeated, because a missing
    hand-over would show as a rare difference."""
    from dvd_gan_amd import kern as K
    F_, S, Cin, Cout, k, up2, relu, ns = case
    torch.manual_seed(F_ + S + Cin + Cout)
    Sin = S // 2 if up2 else S
    x = torch.randn(F_, Sin, Sin, K.pad8(Cin), device=DEV).to(torch.bfloat16)
    if K.pad8(Cin) != Cin:
        x[..., Cin:] = 0
    pk = K.PackedConv(torch.bfloat16, Cout, Cin, (k, k), DEV).fill(torch.randn(Cout, Cin, k, k, device=DEV) * 0.05)
    wq = pk.fragment_major("wf")
    kw = dict(up2=bool(up2), relu_in=bool(relu), nsplit=ns, slabs=ns > 1)
    assert K.wants_fragment_major(torch.bfloat16, F_, S, S, K.pad8(Cin), Cout, k, ns) or up2
    ref = K.conv_forward(x, pk.wf, (k, k), Cout, **kw).clone()
    for _ in range(5):
        assert torch.equal(ref, K.conv_forward(x, pk.wf, (k, k), Cout, wq=wq, **kw))


def test_weights_from_l2_kernel_3d_taps():
    from dvd_gan_amd import kern as K
    x = torch.randn(16, 12, 32, 32, 64, device=DEV).to(torch.bfloat16)
    pk = K.PackedConv(torch.bfloat16, 64, 64, (3, 3, 3), DEV).fill(torch.randn(64, 64, 3, 3, 3, device=DEV) * 0.05)
    ref = K.conv_forward(x, pk.wf, (3, 3, 3), 64).clone()
    assert torch.equal(ref, K.conv_forward(x, pk.wf, (3, 3, 3), 64, wq=pk.fragment_major("wf")))


@pytest.mark.parametrize("case", [(768, 128, 64, 64, True), (768, 256, 128, 32, False), (3072, 256, 256, 16, True)])
def test_upsampled_weight_gradient_on_the_planners_own_split(case):
    """The weight gradient of a 3 x 3 convolution over a nearest-x2 upsampled input (GResBlock.py:57-58) at sizes where the planner
    itself takes the folded form (round 6: conv_wgrad_row4_kernel<.., FOLD> on the input grid, four dy phases as output channels;
    the CASES above force it with msplit) -- generator stages 32 -> 64, 16 -> 32 and 8 -> 16 pixels at a quarter / the full batch:
    against an fp64 restatement of the same sums on the bf16-ROUNDED operands (only the kernels' fp32 summation order differs: 1e-5
    as for the other sliced checks), weight and bias gradient, accumulate and overwrite mode."""
    from dvd_gan_amd import kern as K
    F_, Cin, Cout, S, relu_in = case
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1234 + S)
    x = bf(torch.randn(F_, Cin, S // 2, S // 2, device=dev, generator=g))
    gy = bf(torch.randn(F_, Cout, S, S, device=dev, generator=g) * 0.1)
    # fp64 reference, tap by tap, in chunks of frames: dw[co][ci][ky][kx] = sum over (f, y, x) of gy[f][co][y][x] * xin[f][ci][y + ky - 1][x + kx - 1]
    dw_ref = torch.zeros(Cout, Cin, 3, 3, device=dev, dtype=torch.float64)
    db_ref = gy.double().sum((0, 2, 3))
    for f0 in range(0, F_, 64):
        xin = F.interpolate(F.relu(x[f0:f0 + 64]) if relu_in else x[f0:f0 + 64], scale_factor=2)
        xin = F.pad(xin, (1, 1, 1, 1)).double()
        gyc_ = gy[f0:f0 + 64].double().permute(1, 0, 2, 3).reshape(Cout, -1)
        for ky in range(3):
            for kx in range(3):
                win = xin[:, :, ky:ky + S, kx:kx + S].permute(1, 0, 2, 3).reshape(Cin, -1)
                dw_ref[:, :, ky, kx] += gyc_ @ win.t()
    xc, gyc = K.to_cl(x, torch.bfloat16), K.to_cl(gy, torch.bfloat16)
    dw, db = torch.zeros(Cout, Cin, 3, 3, device=dev), torch.zeros(Cout, device=dev)
    K.conv_wgrad(xc, gyc, dw, (3, 3), Cout, Cin, up2=True, relu_in=relu_in, dbias=db)
    assert rel(dw, dw_ref) < 1e-5
    assert rel(db, db_ref) < 2e-5      # (3.1 M - 12.5 M fp32 additions per channel)
    dw2 = torch.full_like(dw, 7.0)
    K.conv_wgrad(xc, gyc, dw2, (3, 3), Cout, Cin, up2=True, relu_in=relu_in, overwrite=True)
    assert torch.equal(dw2, dw)                          # fixed-order reduction: the same bits, written instead of added to zeros
    K.conv_wgrad(xc, gyc, dw2, (3, 3), Cout, Cin, up2=True, relu_in=relu_in)
    assert rel(dw2, 2 * dw_ref) < 1e-5
