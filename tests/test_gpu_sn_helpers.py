"""GPU parity of the spectral-norm kernels (F1) and the frame helpers (F8) against the reference fixtures."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.detach().double().cpu()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("case", ["conv2d", "conv3d", "linear", "embed"])
def test_f1_spectral_norm_kernels(golden, case):
    """Normalization.py:19-31 / 61-63: one power iteration per forward (u, v in place), sigma = u.W v, W / sigma; the
    gradient wrt weight_bar treats u and v as constants but uses the LATEST u / v (quirk 7)."""
    from dvd_gan_amd import kern as K
    g = sub(golden("f1_spectral_norm"), case)
    w_bar = torch.as_tensor(g["sd0.module.weight_bar"]).to(DEV)
    u = torch.as_tensor(g["sd0.module.weight_u"]).to(DEV)
    v = torch.as_tensor(g["sd0.module.weight_v"]).to(DEV)
    sigma = K.sn_power_iter(w_bar, u, v)
    assert rel(w_bar / sigma, g["out.w1"]) < 2e-6
    assert rel(u, g["sd1.module.weight_u"]) < 2e-6 and rel(v, g["sd1.module.weight_v"]) < 2e-6
    # dL/d(W/sigma) from a plain torch op on the fixture's normalised weight, then the SN backward kernel
    w1 = torch.as_tensor(g["out.w1"]).clone().requires_grad_(True)
    x, gy = torch.as_tensor(g["in.x"]), torch.as_tensor(g["in.gy"])
    if case == "conv2d":
        y = F.conv2d(x, w1, None, padding=1)
    elif case == "conv3d":
        y = F.conv3d(x, w1, None, padding=1)
    elif case == "linear":
        y = F.linear(x, w1)
    else:
        y = F.embedding(x, w1)
    y.backward(gy)
    dW = K.sn_backward(w1.grad.to(DEV).contiguous(), w_bar, u, v, sigma)
    assert rel(dW, g["grad.module.weight_bar"]) < 5e-6
    K.sn_power_iter(w_bar, u, v)
    K.sn_power_iter(w_bar, u, v)
    assert rel(u, g["sd3.module.weight_u"]) < 5e-6 and rel(v, g["sd3.module.weight_v"]) < 5e-6


def test_f8_helpers(golden):
    """utils.py:60-63 / 77-83 on the HIP path, forward bit-exact gather, pooled copy and its gradient."""
    from dvd_gan_amd.helpers import sample_k_frames, vid_downsample
    from oracle import dvdgan_cpu as O
    g = golden("f8_helpers")
    data = torch.as_tensor(g["in.data"]).to(DEV)
    ids4 = O.frame_ids_from_perm(g["in.perm"], 4)
    ids9 = O.frame_ids_from_perm(g["in.perm_k9"], 9)                 # k > T: every frame, sorted
    assert torch.equal(sample_k_frames(data, 6, 4, ids4).cpu(), torch.as_tensor(g["out.sample_k4"]))
    assert torch.equal(sample_k_frames(data, 6, 9, ids9).cpu(), torch.as_tensor(g["out.sample_k9"]))
    xg = data.clone().requires_grad_(True)
    down = vid_downsample(xg)
    np.testing.assert_allclose(down.detach().cpu().numpy(), g["out.down"], rtol=1e-6, atol=1e-7)
    gy = torch.randn(down.shape, generator=torch.Generator().manual_seed(2))
    down.backward(gy.to(DEV))
    xr = torch.as_tensor(g["in.data"]).clone().requires_grad_(True)
    B, T, C, H, W = xr.shape
    want = F.avg_pool2d(xr.view(B * T, C, H, W), 2).view(B, T, C, H // 2, W // 2).permute(0, 2, 1, 3, 4)
    want.backward(gy)
    assert rel(xg.grad, xr.grad) < 1e-6
    # gradient of the gather: scattered back to the sampled frames only
    xs = data.clone().requires_grad_(True)
    s = sample_k_frames(xs, 6, 4, ids4)
    s.backward(torch.ones_like(s))
    mask = torch.zeros(6)
    mask[torch.as_tensor(ids4)] = 1
    assert torch.equal(xs.grad.cpu(), mask.view(1, 6, 1, 1, 1).expand_as(xs.grad.cpu()).contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cbn_backward_relu_mask_from_x_equals_the_stored_activation(dtype):
    """The CBN backward kernels do not read the stored activation: they re-evaluate gamma * xhat + beta from x (the same fused
    multiply-add as the forward, `cbn_affine`) and mask with its sign.  Checked where it could go wrong: beta is chosen so that
    the pre-activation of MANY elements sits within a few ulps of zero, and the result is compared with a torch evaluation that
    masks with the activation the forward kernel actually stored.  A single element masked differently would show up as a
    gradient error of the size of that element's incoming gradient (O(1))."""
    from dvd_gan_amd import kern as K
    torch.manual_seed(3)
    F_, H, W, C, B = 24, 8, 8, 16, 4
    x = torch.randn(F_, H, W, C, device=DEV).to(dtype)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, rstd = K.bn_stats(x, C, True, 1e-5, 0.1, rm, rv)
    # statistics through the 16 copies of the sums: against torch in fp64
    xd = x.double().reshape(-1, C)
    assert rel(mean, xd.mean(0).cpu()) < 1e-6 and rel(rstd, (1.0 / (xd.var(0, unbiased=False) + 1e-5).sqrt()).cpu()) < 1e-6
    samp = (torch.arange(F_, device=DEV, dtype=torch.int32) % B).contiguous()
    gamma = torch.randn(B, C, device=DEV)
    xhat = (x.float() - mean) * rstd
    # per (condition row, channel): beta = -gamma * xhat of one pixel of a frame with that condition -> that pixel's t == +-0 or a
    # few ulps, and every other pixel's t is an ordinary number
    beta = torch.empty(B, C, device=DEV)
    for s in range(B):
        beta[s] = -(gamma[s] * xhat[s, 3, 5])
    gb = torch.cat([gamma, beta], 1).contiguous()
    a = K.cbn_apply(x, C, mean, rstd, gb, samp, True)
    g = torch.randn(F_, H, W, C, device=DEV).to(dtype)
    dx, dgb = K.cbn_backward(g, None, x, C, mean, rstd, gb, samp, True)
    # torch evaluation with the STORED activation as the mask
    mask = (a.float() > 0).float()
    gm = g.float() * mask
    gam_f = gamma[samp.long()].view(F_, 1, 1, C)
    dgamma = torch.zeros(B, C, device=DEV).index_add_(0, samp.long(), (gm * xhat).sum((1, 2)))
    dbeta = torch.zeros(B, C, device=DEV).index_add_(0, samp.long(), gm.sum((1, 2)))
    dxhat = gm * gam_f
    n = F_ * H * W
    dx_ref = rstd * (dxhat - dxhat.sum((0, 1, 2)) / n - xhat * (dxhat * xhat).sum((0, 1, 2)) / n)
    near = int(((gam_f * xhat + beta[samp.long()].view(F_, 1, 1, C)).abs() < 1e-5).sum())
    assert near >= B * C                                     # the adversarial elements are really there
    tol = 1e-5 if dtype == torch.float32 else 8e-3           # bf16: rounding of the stored dx only
    assert float((dx.float() - dx_ref).abs().max()) < tol * float(dx_ref.abs().max() + 1)
    assert rel(dgb[:, :C], dgamma.cpu()) < 1e-5 and rel(dgb[:, C:], dbeta.cpu()) < 1e-5
