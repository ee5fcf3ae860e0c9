"""The import-swap scenario of INTEGRATION.md section 1: the three HIP-backed modules and the two helpers driven by code
shaped like the reference's own loop (trainer.py:223-307) -- `.cuda()`, STOCK torch.optim.Adam on the requires_grad
parameters, StepLR, `zero_grad()` (set-to-none) on all three optimizers before every backward, no FlatAdam, no frozen
discriminator weights in the generator step -- for the two steps of golden F9 (produced by the unmodified reference
Trainer).  Losses and post-step parameter checksums must match the reference like the native Trainer's do."""
import numpy as np
import pytest
import torch
from torch.optim.lr_scheduler import StepLR

from conftest import sub

pytestmark = pytest.mark.gpu


def calc_loss(x, real_flag):                 # trainer.py:114-121 (hinge), plain torch ops on the [B*k] logits
    return torch.nn.functional.relu(1.0 - x).mean() if real_flag else torch.nn.functional.relu(1.0 + x).mean()


def test_reference_loop_with_stock_adam(golden):
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    from dvd_gan_amd.gen_net import Generator
    from dvd_gan_amd.helpers import sample_k_frames, vid_downsample
    g = golden("f9_trainer_hinge")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    lr = float(g["meta.lr"])
    f32 = torch.float32
    G = Generator(z_dim, 4, n_class, ch, T, compute_dtype=f32)
    D_s, D_t = SpatialDiscriminator(ch, n_class, compute_dtype=f32), TemporalDiscriminator(ch, n_class, compute_dtype=f32)
    for net, tag in ((G, "G"), (D_s, "Ds"), (D_t, "Dt")):
        net.load_state_dict({kk: torch.as_tensor(v) for kk, v in sub(g, tag + ".sd0").items()})
    G, D_s, D_t = G.cuda(), D_s.cuda(), D_t.cuda()                       # trainer.py:349-351
    opts = [torch.optim.Adam(filter(lambda p: p.requires_grad, n.parameters()), lr, (0.0, 0.9)) for n in (G, D_s, D_t)]
    g_opt, ds_opt, dt_opt = opts
    scheds = [StepLR(o, step_size=10000, gamma=1) for o in opts]         # lr_schr='const', trainer.py:142-145

    def reset_grad():                                                    # trainer.py:384-387
        ds_opt.zero_grad(); dt_opt.zero_grad(); g_opt.zero_grad()
    G.train(); D_s.train(); D_t.train()
    for s in range(steps):
        real_videos = torch.as_tensor(g[f"in.real.{s}"]).cuda().permute(0, 2, 1, 3, 4).contiguous()
        real_labels = torch.as_tensor(g[f"in.labels.{s}"]).cuda()
        ids_real = torch.as_tensor(g[f"in.perm_real.{s}"])[:k].sort()[0]  # the reference's randperm draws (utils.py:61-62)
        ids_fake = torch.as_tensor(g[f"in.perm_fake.{s}"])[:k].sort()[0]
        real_s = sample_k_frames(real_videos, T, k, ids_real)
        z, z_class = torch.as_tensor(g[f"in.z.{s}"]).cuda(), torch.as_tensor(g[f"in.z_class.{s}"]).cuda()
        fake = G(z, z_class)
        fake_s = sample_k_frames(fake, T, k, ids_fake)
        ds_real, ds_fake = calc_loss(D_s(real_s, real_labels), True), calc_loss(D_s(fake_s.detach(), z_class), False)
        reset_grad()
        (ds_real + ds_fake).backward()
        ds_opt.step(); scheds[1].step()
        real_d, fake_d = vid_downsample(real_videos), vid_downsample(fake)
        dt_real, dt_fake = calc_loss(D_t(real_d, real_labels), True), calc_loss(D_t(fake_d.detach(), z_class), False)
        reset_grad()
        (dt_real + dt_fake).backward()
        dt_opt.step(); scheds[2].step()
        g_s, g_t = calc_loss(D_s(fake_s, z_class), True), calc_loss(D_t(fake_d, z_class), True)
        reset_grad()
        (g_s + g_t).backward()                       # also fills the (unused) discriminator gradients, like the reference
        assert all(p.grad is not None for p in D_s.parameters() if p.requires_grad)
        g_opt.step(); scheds[0].step()
        got = [float(v) for v in (ds_real, ds_fake, dt_real, dt_fake, g_s, g_t)]
        want = g[f"out.losses.{s}"]
        if s == 0:
            np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
        else:                                        # one Adam update deep: see tests/test_gpu_trainer.py
            np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-2)
        keys = [str(x) for x in g["meta.psum_keys.G"]]
        sd = G.state_dict()
        psum = np.array([float(sd[kk].double().abs().sum()) for kk in keys])
        ref = g[f"out.psum.{s}.G"]
        big = ref > 1.0
        np.testing.assert_allclose(psum[big], ref[big], rtol=1e-3 if s == 0 else 2e-2)


@pytest.mark.parametrize("after_failed_backward", [False, True])
def test_persistent_grads_are_joined_when_backward_returns(golden, after_failed_backward):
    """once a Trainer has switched the direct weight-gradient route on, ANY backward() over parameters whose .grad
    is a persistent fp32 buffer (stock optimizer + zero_grad(set_to_none=False), gradient accumulation, a tool driving
    the generator alone) accumulates on the side stream.  The join is queued as a final callback of the autograd engine:
    when backward() returns the calling stream already waits for it, so reading / stepping right away is race-free.
    Checked against the autograd route on the same generator: identical gradients, twice in a row (accumulation).
    after_failed_backward: a backward() that RAISES half-way (a user hook error, an OOM that is caught and retried)
    makes the engine drop its final callbacks; the join of every later backward must still happen."""
    from dvd_gan_amd import functional as Fn
    from dvd_gan_amd.gen_net import Generator
    g = golden("f9_trainer_hinge")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    G = Generator(z_dim, 4, n_class, ch, T, compute_dtype=torch.float32)
    G.load_state_dict({kk: torch.as_tensor(v) for kk, v in sub(g, "G.sd0").items()})
    G = G.cuda().train()
    sd0 = {kk: v.clone() for kk, v in G.state_dict().items()}
    z, zc = torch.as_tensor(g["in.z.0"]).cuda(), torch.as_tensor(g["in.z_class.0"]).cuda()
    w = torch.randn(B, T, 3, 64, 64, device="cuda")

    class Boom(RuntimeError):
        pass

    def two_backwards(direct):
        G.load_state_dict(sd0)                                  # same SN u / v, BN statistics for both routes
        Fn.direct_weight_grads(direct)
        for p in G.parameters():
            p.grad = torch.zeros_like(p) if (direct and p.requires_grad) else None    # persistent buffers <-> set-to-none
        if after_failed_backward and direct:
            # a backward pass that has queued side-stream work for nearly every layer and then dies in a tensor hook on the
            # embedding weight, whose gradient is produced at the very end of the pass
            hit = []

            def bad_hook(grad):
                hit.append(1)
                raise Boom("hook error in the middle of backward")
            bad = G.embedding.weight.register_hook(bad_hook)
            with pytest.raises(Boom):
                (G(z, zc) * w).sum().backward()
            bad.remove()
            assert hit
            torch.cuda.synchronize()
            G.load_state_dict(sd0)                              # the failed pass advanced SN u / v and the BN statistics
            for p in G.parameters():
                if p.grad is not None:
                    p.grad.zero_()
        for _ in range(2):                                      # second pass accumulates
            (G(z, zc) * w).sum().backward()
        # no join_side() here on purpose: the gradients are read straight after backward()
        return {kk: p.grad.detach().clone() for kk, p in G.named_parameters() if p.grad is not None}
    try:
        ref = two_backwards(False)
        got = two_backwards(True)
        assert Fn._SIDE["stream"] is not None                   # the side stream really was used
    finally:
        Fn.direct_weight_grads(False)
    assert set(got) == set(ref)
    scale = max(float(v.norm()) for v in ref.values())
    for kk in ref:
        # (a bias in front of a batch norm has a mathematically zero gradient: rounding noise in both routes, hence the
        #  absolute term)
        d = float((got[kk] - ref[kk]).norm())
        assert d < 1e-5 * float(ref[kk].norm()) + 1e-6 * scale, (kk, d, float(ref[kk].norm()))
