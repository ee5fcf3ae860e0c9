"""GPU parity of the full G + D_s + D_t training step (HIP path) against the unmodified reference
Trainer (golden F9: two steps, hinge and wgan-gp; F10: config-1 plumbing, three steps).

Exact mode (f32 MFMA): the six loss terms per step within 2e-3 relative / 2e-4 absolute of the
reference (two optimizer steps deep; the fixtures amplify fp32 rounding ~100x, see
test_gpu_modules.GEN_TOL), post-step parameter checksums within 3e-2 (see comment), named gradients rel-L2 < 2e-2.
bf16 mode: F9 losses within 2e-2 (step 0) / 5e-2 (step 1, after an Adam step at 40x the reference lr) absolute; F14 (one
step from the reference's OWN default initialisation, bf16-representable state): losses within 1e-2,
gradients judged against the one-ulp sensitivity of the exact mode; 20-step bf16-vs-exact trajectory within 1e-2.
"""
import argparse

import numpy as np
import pytest
import torch

from conftest import fixture_real, full_states, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_trainer(g, base, adv, dtype):
    from dvd_gan_amd.train_step import Trainer
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    lr = float(g["meta.lr"])
    cfg = argparse.Namespace(adv_loss=adv, z_dim=z_dim, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=int(g["meta.d_iters"]) if "meta.d_iters" in g else 1, batch_size=B,
                             g_lr=lr, d_lr=lr, beta1=0.0, beta2=0.9, n_class=n_class, k_sample=k)
    tr = Trainer([], cfg, device=torch.device(DEV), compute_dtype=dtype)
    for net, sd in zip((tr.G, tr.D_s, tr.D_t), full_states(base)):
        net.load_state_dict({kk: torch.as_tensor(v) for kk, v in sd.items()})
        net.train()
    return tr, steps


def rel_l2(a, b):
    b = torch.as_tensor(b).double()
    return float((a.detach().double().cpu() - b).norm() / (b.norm() + 1e-30))


def check_step0_grads(tr, g):
    """Named gradients the reference held when each optimizer stepped in step 0 (no Adam involved
    yet): D_s / D_t gradients of their own updates, G gradients through the UPDATED discriminators."""
    n = 0
    for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
        named = dict(net.named_parameters())
        for k, v in sub(g, f"grad.0.{tag}").items():
            r = rel_l2(tr._grad_snap[tag][k], v)
            assert r < 1e-2, (tag, k, r)
            n += 1
    assert n >= 10


def run(g, base, adv, dtype):
    tr, steps = make_trainer(g, base, adv, dtype)
    # snapshot gradients right before each Adam launch of step 0
    tr._grad_snap = {}
    for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
        def wrap(opt=opt, net=net, tag=tag, orig=opt.step):
            def stepper():
                if tag not in tr._grad_snap:
                    tr._grad_snap[tag] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
                orig()
            return stepper
        opt.step = wrap()
    nb = len([k for k in base if k.startswith("in.labels.")])
    out = []
    for s in range(steps):
        draws = {"perm_real": base[f"in.perm_real.{s}"], "z": base[f"in.z.{s}"], "z_class": base[f"in.z_class.{s}"],
                 "perm_fake": base[f"in.perm_fake.{s}"]}
        losses = tr.train_step(torch.as_tensor(fixture_real(base, s % nb)), torch.as_tensor(base[f"in.labels.{s % nb}"]), draws)
        out.append([float(v.detach()) for v in losses])
        want = g[f"out.losses.{s}"]
        if dtype == torch.float32:
            # step 0 is a pure forward/backward comparison.  From step 1 on the weights have been
            # through Adam, whose first updates are ~lr*sign(g): elements whose gradient is at the
            # rounding-noise level flip freely.  Measured with the CPU oracle on this fixture (fp64
            # vs fp32, and fp64 with 1e-7 relative weight noise) that moves the step-1 losses by
            # ~2e-3; the HIP path reorders more sums than that, so later steps get atol 2e-2.
            if s == 0:
                np.testing.assert_allclose(out[-1], want, rtol=2e-3, atol=2e-4, err_msg="losses step 0")
            else:
                np.testing.assert_allclose(out[-1], want, rtol=1e-2, atol=2e-2, err_msg=f"losses step {s}")
            keys = [str(x) for x in g["meta.psum_keys.G"]]
            sd = tr.G.state_dict()
            got = np.array([float(sd[kk].double().abs().sum()) for kk in keys])
            ref = g[f"out.psum.{s}.G"]
            big = ref > 1.0                  # zero-initialised biases are sums of +-lr noise steps
            np.testing.assert_allclose(got[big], ref[big], rtol=1e-3 if s == 0 else 2e-2, err_msg=f"G checksums step {s}")
        else:
            # bf16 on F9 (ch=2, weights NOT bf16-representable, lr 2e-3 = 40x the reference's): measured 8.1e-3 at step 0 and
            # 1.9e-2 one (large) Adam step later.  The tight statement of the timed mode -- 1e-2 at the reference's own
            # initialisation, judged against the one-ulp sensitivity -- is test_step_at_reference_default_init (F14).
            np.testing.assert_allclose(out[-1], want, atol=2e-2 if s == 0 else 5e-2, err_msg=f"losses step {s}")
    return tr, out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainer_hinge_two_steps(golden, dtype):
    g = golden("f9_trainer_hinge")
    tr, _ = run(g, g, "hinge", dtype)
    if dtype == torch.float32:
        check_step0_grads(tr, g)
        # (gradients of the SECOND generator update are not compared: on this fixture the CPU oracle
        #  run in fp64 differs from its own fp32 run by 70-130 % there -- chaotic after one Adam step)
        for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):      # SN u/v, BN buffers after 2 steps
            sd = net.state_dict()
            for k, v in sub(g, tag + ".sd1").items():
                if not k.endswith("num_batches_tracked"):
                    assert float((sd[k].double().cpu() - torch.as_tensor(v).double()).norm() / (np.linalg.norm(v) + 1e-30)) < 5e-2, k


def test_trainer_wgangp_two_steps(golden):
    g = golden("f9_trainer_wgangp")
    tr, _ = run(g, golden("f9_trainer_hinge"), "wgan-gp", torch.float32)
    check_step0_grads(tr, g)


def test_config1_plumbing_three_steps(golden):
    g = golden("f10_config1")
    run(g, g, "hinge", torch.float32)


def test_adam_kernel_matches_torch():
    """dvd_adam_step against torch.optim.Adam (CPU) on random data, 3 steps, betas of the reference."""
    from dvd_gan_amd import kern as K
    torch.manual_seed(3)
    p0 = torch.randn(10007)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-3, betas=(0.0, 0.9))
    p = p0.to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(10007) * 10 ** float(torch.randint(-6, 1, ()))
        ref.grad = gr.clone()
        opt.step()
        K.adam_step(p, gr.to(DEV), m, v, 2e-3, 0.0, 0.9, 1e-8, step)
        assert float((p.cpu() - ref.detach()).abs().max()) < 1e-6


def test_step_at_128x128_frames_matches_oracle():
    """BASELINE configs[3] frame size (latent_dim = 8 -> 128 x 128 clips): one hinge step at ch=2, T=8, k=4, B=2 in
    exact mode against the CPU oracle on the same weights, inputs and RNG draws.  Exercises the 128-wide GResBlock /
    colorize convs, D_s attention at 1024 tokens (8-row attention blocks) and D_t on 64 x 64 pooled frames."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.train_step import Trainer
    torch.manual_seed(13)
    ch, T, k, B, ncls, zd, ld = 2, 8, 4, 2, 3, 16, 8
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=zd, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=ncls, k_sample=k)
    tr = Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.float32, latent_dim=ld)
    sds = [O.make_state({kk: v.detach().cpu().clone() for kk, v in net.state_dict().items()})
           for net in (tr.G, tr.D_s, tr.D_t)]
    st = O.TrainState(*sds, ch=ch, n_frames=T, k_sample=k, n_class=ncls, z_dim=zd, latent_dim=ld)
    real = torch.rand(B, 3, T, 128, 128) * 2 - 1
    labels = torch.randint(0, ncls, (B,))
    draws = {"perm_real": torch.randperm(T), "z": torch.randn(B, zd), "z_class": torch.randint(0, ncls, (B,)),
             "perm_fake": torch.randperm(T)}
    got = [float(v.detach()) for v in tr.train_step(real, labels, draws)]
    want = O.train_step(st, real, labels, draws["z"], draws["z_class"], draws["perm_real"], draws["perm_fake"])
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)


def test_sampling_path_matches_oracle():
    """trainer.py:323-334: eval-mode generator on fixed z / labels (BN running statistics) + utils.denorm, checked against
    the oracle's eval-mode generator after one training step has moved the running statistics."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.train_step import Trainer
    torch.manual_seed(21)
    ch, T, k, B, ncls, zd = 2, 8, 4, 2, 3, 16
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=zd, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=ncls, k_sample=k)
    tr = Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.float32)
    tr.train_step(torch.rand(B, 3, T, 64, 64) * 2 - 1, torch.randint(0, ncls, (B,)))
    sd = O.make_state({kk: v.detach().cpu().clone() for kk, v in tr.G.state_dict().items()}, requires_grad=False)
    fixed_z, fixed_label = torch.randn(B, zd), torch.randint(0, ncls, (B,))
    with torch.no_grad():
        want = ((O.generator(sd, fixed_z, fixed_label, ch, T, training=False) + 1) / 2).clamp(0, 1)
    got = tr.sample(fixed_z, fixed_label)
    assert tr.G.training
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    assert float((got.cpu() - want).abs().max()) < 2e-3


def test_step_with_padded_channel_counts_matches_oracle():
    """ch=6 (48 / 24 / 12 generator channels, 12..96 discriminator channels: 12 is no multiple of 8 -> zero-padded channel
    vectors in colorize and the first discriminator stages, attention with THREE query channels), B=3 with T=8 (B does not divide T: the condition mis-ordering wraps), k_sample larger than
    T (every frame goes to D_s, utils.py:61-62): one exact-mode hinge step against the oracle."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.train_step import Trainer
    torch.manual_seed(31)
    ch, T, k, B, ncls, zd = 6, 8, 11, 3, 4, 10
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=zd, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=ncls, k_sample=k)
    tr = Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.float32)
    sds = [O.make_state({kk: v.detach().cpu().clone() for kk, v in net.state_dict().items()})
           for net in (tr.G, tr.D_s, tr.D_t)]
    st = O.TrainState(*sds, ch=ch, n_frames=T, k_sample=k, n_class=ncls, z_dim=zd)
    real = torch.rand(B, 3, T, 64, 64) * 2 - 1
    labels = torch.randint(0, ncls, (B,))
    draws = {"perm_real": torch.randperm(T), "z": torch.randn(B, zd), "z_class": torch.randint(0, ncls, (B,)),
             "perm_fake": torch.randperm(T)}
    got = [float(v.detach()) for v in tr.train_step(real, labels, draws)]
    want = O.train_step(st, real, labels, draws["z"], draws["z_class"], draws["perm_real"], draws["perm_fake"])
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
    with pytest.raises(IndexError):
        tr.train_step(real, torch.full((B,), ncls), draws)          # label == n_class: rejected on the host


def test_two_discriminator_iterations_golden(golden):
    """Golden F13: the reference Trainer with d_iters = 2 (trainer.py:230), B = 1, two steps.  Exact mode: step-0 losses (last
    discriminator iteration + generator) at the tolerance of the F9 test, later step looser (one Adam update deep), and the
    forward-only state (BN running statistics, spectral-norm vectors: advanced 2 x 2 times) after both steps."""
    g = golden("f13_two_d_iters")
    tr, steps = make_trainer(g, g, "hinge", torch.float32)
    assert tr.d_iters == 2
    for s in range(steps):
        draws = [{"perm_real": g[f"in.perm_real.{s}.{i}"], "z": g[f"in.z.{s}.{i}"], "z_class": g[f"in.z_class.{s}.{i}"],
                  "perm_fake": g[f"in.perm_fake.{s}.{i}"]} for i in range(2)]
        got = [float(v.detach()) for v in tr.train_step(torch.as_tensor(fixture_real(g, s)), torch.as_tensor(g[f"in.labels.{s}"]), draws)]
        if s == 0:
            np.testing.assert_allclose(got, g[f"out.losses.{s}"], rtol=2e-3, atol=2e-4, err_msg="losses step 0")
        else:
            np.testing.assert_allclose(got, g[f"out.losses.{s}"], rtol=1e-2, atol=2e-2, err_msg=f"losses step {s}")
    for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):
        sd = net.state_dict()
        for k, v in sub(g, tag + ".sd1").items():
            if k.endswith(("weight_u", "weight_v", "running_mean", "running_var")):
                assert float((sd[k].double().cpu() - torch.as_tensor(v).double()).norm() / (np.linalg.norm(v) + 1e-30)) < 5e-2, k


def test_two_discriminator_iterations_per_step_match_oracle():
    """d_iters = 2 (trainer.py:230): two discriminator updates with fresh z / labels / frame draws each, generator forward in
    train mode both times (BN running statistics and spectral-norm vectors advance twice), then the generator step on the clips
    of the LAST iteration.  B = 1: the smallest batch (statistics over the T frames of one clip).  Two steps, exact mode."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.train_step import Trainer
    torch.manual_seed(41)
    ch, T, k, B, ncls, zd = 2, 8, 4, 1, 3, 12
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=zd, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=2, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=ncls, k_sample=k)
    tr = Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.float32)
    sds = [O.make_state({kk: v.detach().cpu().clone() for kk, v in net.state_dict().items()})
           for net in (tr.G, tr.D_s, tr.D_t)]
    st = O.TrainState(*sds, ch=ch, n_frames=T, k_sample=k, n_class=ncls, z_dim=zd)
    for step in range(2):
        real = torch.rand(B, 3, T, 64, 64) * 2 - 1
        labels = torch.randint(0, ncls, (B,))
        draws = [{"perm_real": torch.randperm(T), "z": torch.randn(B, zd), "z_class": torch.randint(0, ncls, (B,)),
                  "perm_fake": torch.randperm(T)} for _ in range(2)]
        got = [float(v.detach()) for v in tr.train_step(real, labels, draws)]
        want = O.train_step(st, real, labels, [d["z"] for d in draws], [d["z_class"] for d in draws],
                            [d["perm_real"] for d in draws], [d["perm_fake"] for d in draws], d_iters=2)
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
    # the generator state that only moves in forward passes advanced 2 x 2 times on both sides
    sd = tr.G.state_dict()
    for key in ("conv.1.CBNorm1.bn.running_mean", "conv.1.CBNorm2.bn.running_var", "colorize.module.weight_u"):
        assert float((sd[key].cpu() - st.G[key].detach()).abs().max()) < 2e-4, key


def test_temporal_discriminator_rejects_frame_counts_it_cannot_pool():
    from dvd_gan_amd.disc_nets import TemporalDiscriminator
    D = TemporalDiscriminator(2, 3, compute_dtype=torch.float32).to(DEV)
    with pytest.raises(ValueError, match="multiple of 4"):
        D(torch.zeros(1, 3, 6, 32, 32, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))


# ------------------------------------------------------------------ bf16 at the reference's own initialisation (F14)
def _snap_step(tr, g, s=0):
    """One train_step on the fixture's draws with the gradients captured right before each Adam launch."""
    snap = {}
    for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
        def wrap(opt=opt, net=net, tag=tag, orig=opt.step):
            def stepper():
                if tag not in snap:
                    snap[tag] = {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}
                orig()
            return stepper
        opt.step = wrap()
    draws = {"perm_real": g[f"in.perm_real.{s}"], "z": g[f"in.z.{s}"], "z_class": g[f"in.z_class.{s}"],
             "perm_fake": g[f"in.perm_fake.{s}"]}
    losses = tr.train_step(torch.as_tensor(fixture_real(g, s)), torch.as_tensor(g[f"in.labels.{s}"]), draws)
    return [float(v.detach()) for v in losses], snap


def _cos(a, b):
    """cosine of two gradients; 1.0 when the reference gradient is exactly zero and so is ours (attention projections
    behind gamma = 0, Discriminators.py:91), 0.0 when only one of them is"""
    a, b = a.double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    if float(b.norm()) == 0.0:
        return 1.0 if float(a.norm()) == 0.0 else 0.0
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


F14_NUMBERS = {}


@pytest.mark.parametrize("mode", ["exact", "bf16", "ulp"])
def test_step_at_reference_default_init(golden, mode):
    """ONE step from the reference's own default initialisation (orthogonal ConvGRU weights: the state a fresh model starts
    from), all weights / clips bf16-representable so the bf16 mode sees exactly the reference's operands (fixture F14,
    ch=4, T=16, B=2).  Three runs against the reference's recorded step:
      exact : fp32 path.  Six losses 2e-3 rel / 2e-4 abs (measured 2e-7); |grad| checksum of every parameter 1e-2 rel
              (measured 9e-4); named gradients cosine >= 0.9999.
      bf16  : the timed mode.  Six losses within 1e-2 absolute -- SURVEY section 8c's bound (measured 1.5e-3); D_s / D_t:
              checksum vector rel-L2 <= 1e-2 (measured 1.7e-3), named gradients cosine >= 0.999 (measured 0.9998); G:
              named gradients cosine >= 0.9 (measured 0.937 ... 0.9999: the early layers sit behind all four recurrences).
      ulp   : exact mode with every weight moved by ONE bf16 ulp (relative 2^-9, random sign) -- the least any bf16
              implementation perturbs them.  It yields the sensitivity the bf16 numbers are judged against: asserted below
              that the bf16 mode's loss errors and generator-gradient misalignment (1 - cosine) stay within 3x the
              one-ulp run's (+ 1e-3), i.e. the 0.94 is the conditioning of the recurrence at this initialisation, not
              kernel error."""
    import json, os
    g = golden("f14_default_init_bf16")
    exact = mode != "bf16"
    tr, _ = make_trainer(g, g, "hinge", torch.float32 if exact else torch.bfloat16)
    if mode == "ulp":
        gen = torch.Generator(device="cuda").manual_seed(5)
        with torch.no_grad():
            for net in (tr.G, tr.D_s, tr.D_t):
                for p in net.parameters():
                    if p.requires_grad:
                        sgn = torch.randint(0, 2, p.shape, generator=gen, device=p.device).float() * 2 - 1
                        p.mul_(1 + sgn * 2.0 ** -9)
    losses, snap = _snap_step(tr, g)
    want = g["out.losses.0"]
    rec = {"loss_abs_err": [abs(a - b) for a, b in zip(losses, want)], "losses": losses, "want": [float(v) for v in want]}
    for tag in ("Ds", "Dt", "G"):
        keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
        got = np.array([float(snap[tag][kk].double().abs().sum()) for kk in keys])
        ref = g[f"out.gsum.0.{tag}"]
        rec[f"gsum_relL2_{tag}"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        big = ref > 1e-3 * ref.max()
        rec[f"gsum_maxrel_{tag}"] = float(np.max(np.abs(got[big] - ref[big]) / ref[big]))
        cs = {kk: _cos(snap[tag][kk].reshape(-1)[:v.size], v) for kk, v in sub(g, f"grad.0.{tag}").items()}
        rec[f"cos_min_{tag}"] = min(cs.values())
        rec[f"cos_{tag}"] = cs
    F14_NUMBERS[mode] = rec
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/f14_numbers.json", "w") as f:
        json.dump(F14_NUMBERS, f, indent=1)
    if mode == "exact":
        np.testing.assert_allclose(losses, want, rtol=2e-3, atol=2e-4)
        for tag in ("Ds", "Dt", "G"):
            assert rec[f"gsum_maxrel_{tag}"] < 1e-2, (tag, rec[f"gsum_maxrel_{tag}"])
            assert rec[f"cos_min_{tag}"] > 0.9999, (tag, rec[f"cos_{tag}"])
    elif mode == "bf16":
        np.testing.assert_allclose(losses, want, atol=1e-2, rtol=0)
        for tag in ("Ds", "Dt"):
            assert rec[f"gsum_relL2_{tag}"] < 1e-2, (tag, rec[f"gsum_relL2_{tag}"])
            assert rec[f"cos_min_{tag}"] > 0.999, (tag, rec[f"cos_{tag}"])
        assert rec["cos_min_G"] > 0.9, rec["cos_G"]
    else:
        assert "bf16" in F14_NUMBERS, "runs after the bf16 case"
        b = F14_NUMBERS["bf16"]
        assert max(b["loss_abs_err"]) <= 3 * max(rec["loss_abs_err"]) + 1e-3, (b["loss_abs_err"], rec["loss_abs_err"])
        for kk, c in b["cos_G"].items():
            assert 1 - c <= 3 * (1 - rec["cos_G"][kk]) + 1e-3, (kk, c, rec["cos_G"][kk])


def test_bf16_trajectory_follows_exact_mode(golden):
    """20 optimizer steps from the reference's default initialisation (F14 state, ch=4, T=16, B=2, hinge, lr 5e-5) in exact
    mode and in bf16 mode on the same clips and RNG draws: the bf16 trajectory must stay beside the exact one -- every
    loss term within 2e-2 absolute at every step and 4e-3 on average, nothing non-finite (measured over eight runs: worst
    term 2.9e-3 ... 1.2e-2, mean 1e-3; the runs differ among themselves -- the weight gradients of these 16..32-channel layers go
    through fp32 atomics, so even two exact-mode runs part in the fourth digit after 20 Adam steps; round 3 quoted 1e-2 from a
    single run at 4.1e-3 and the bound then failed about one run in four), and the parameter displacement of the two runs (both have moved every weight by ~20 Adam steps of 5e-5)
    agreeing to cosine >= 0.99 for each network (measured 0.9995 / 0.99999 / 0.99999)."""
    import json, os
    g = golden("f14_default_init_bf16")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    runs = {}
    for dtype in (torch.float32, torch.bfloat16):
        tr, _ = make_trainer(g, g, "hinge", dtype)
        p0 = [torch.cat([p.detach().reshape(-1).cpu() for p in net.parameters()]) for net in (tr.G, tr.D_s, tr.D_t)]
        gen = torch.Generator().manual_seed(77)
        hist = []
        for s in range(20):
            real = (torch.rand(B, 3, T, 64, 64, generator=gen) * 2 - 1).to(torch.bfloat16).float()
            labels = torch.randint(0, n_class, (B,), generator=gen)
            draws = {"perm_real": torch.randperm(T, generator=gen), "z": torch.randn(B, z_dim, generator=gen),
                     "z_class": torch.randint(0, n_class, (B,), generator=gen), "perm_fake": torch.randperm(T, generator=gen)}
            hist.append([float(v.detach()) for v in tr.train_step(real, labels, draws)])
        p1 = [torch.cat([p.detach().reshape(-1).cpu() for p in net.parameters()]) for net in (tr.G, tr.D_s, tr.D_t)]
        runs[dtype] = (np.array(hist), [b - a for a, b in zip(p0, p1)])
    he, hb = runs[torch.float32][0], runs[torch.bfloat16][0]
    assert np.isfinite(he).all() and np.isfinite(hb).all()
    dev = np.abs(he - hb)
    cos = [_cos(a, b) for a, b in zip(runs[torch.float32][1], runs[torch.bfloat16][1])]
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/f14_trajectory.json", "w") as f:
        json.dump({"max_abs_dev_per_term": dev.max(0).tolist(), "max_abs_dev_per_step": dev.max(1).tolist(),
                   "displacement_cosine_G_Ds_Dt": cos, "exact_losses": he.tolist(), "bf16_losses": hb.tolist()}, f, indent=1)
    assert dev.max() < 2e-2, dev.max(0)
    assert dev.mean() < 4e-3, dev.mean(0)
    assert min(cos) > 0.99, cos


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inference_forward_equals_training_forward_and_keeps_no_bptt_tensors(dtype):
    """Trainer.sample() / any generator forward under torch.no_grad() takes the inference form of the ConvGRU layers (u and
    h*r in one-step scratch buffers, r and o never stored): same clips bit for bit as the storing forward, a fraction of
    its memory."""
    from dvd_gan_amd.gen_net import Generator
    torch.manual_seed(5)
    G = Generator(16, 4, 3, 4, 8, compute_dtype=dtype).cuda().eval()
    z, c = torch.randn(4, 16, device="cuda"), torch.randint(0, 3, (4,), device="cuda")
    sd0 = {k: v.clone() for k, v in G.state_dict().items()}

    def run(no_grad):
        G.load_state_dict(sd0)                       # spectral-norm u / v advance with every forward (quirk 2)
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        if no_grad:
            with torch.no_grad():
                y = G(z, c)
        else:
            y = G(z, c)
        torch.cuda.synchronize()
        return y.detach().clone(), torch.cuda.max_memory_allocated() - base
    y_train, mem_train = run(False)
    y_infer, mem_infer = run(True)
    assert torch.equal(y_train, y_infer)
    assert mem_infer < 0.6 * mem_train, (mem_infer, mem_train)


def test_step_at_96x96_frames_matches_oracle():
    """Frames that are NOT powers of two (latent_dim = 6 -> 96 x 96 clips; Discriminators.py:242-291 / 400-447 take any frame
    their poolings can halve): one hinge step at ch=2, T=8, k=4, B=2 in exact mode against the CPU oracle.  Generator and both
    discriminators run the division-indexed tap-by-tap convolution / weight-gradient kernels; D_t's last block pools a 3 x 3 map
    to 1 x 1 with F.avg_pool2d's floor (odd-grid dvd_pool / dvd_unpool); D_s attention over 576 tokens, D_t over 144."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.train_step import Trainer
    torch.manual_seed(17)
    ch, T, k, B, ncls, zd, ld = 2, 8, 4, 2, 3, 16, 6
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=zd, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=ncls, k_sample=k)
    tr = Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.float32, latent_dim=ld)
    sds = [O.make_state({kk: v.detach().cpu().clone() for kk, v in net.state_dict().items()})
           for net in (tr.G, tr.D_s, tr.D_t)]
    st = O.TrainState(*sds, ch=ch, n_frames=T, k_sample=k, n_class=ncls, z_dim=zd, latent_dim=ld)
    snaps_o = O.snapshot_grads(st)
    snaps = {}
    for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
        def stepper(net=net, tag=tag, orig=opt.step):
            snaps[tag] = {kk: p.grad.clone() for kk, p in net.named_parameters() if p.grad is not None}
            orig()
        opt.step = stepper
    real = torch.rand(B, 3, T, 96, 96) * 2 - 1
    labels = torch.randint(0, ncls, (B,))
    draws = {"perm_real": torch.randperm(T), "z": torch.randn(B, zd), "z_class": torch.randint(0, ncls, (B,)),
             "perm_fake": torch.randperm(T)}
    got = [float(v.detach()) for v in tr.train_step(real, labels, draws)]
    want = O.train_step(st, real, labels, draws["z"], draws["z_class"], draws["perm_real"], draws["perm_fake"])
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
    for tag in ("Ds", "Dt", "G"):                 # every parameter gradient of the three updates
        scale = max(float(v.norm()) for v in snaps_o[tag].values())
        for kk, v in snaps_o[tag].items():
            d = float((snaps[tag][kk].double().cpu() - v.double()).norm())
            assert d < 1e-2 * float(v.norm()) + 1e-5 * scale, (tag, kk, d, float(v.norm()))


@pytest.mark.parametrize("size", [80, 72])
def test_discriminators_on_even_frames_that_floor(size):
    """80 x 80 (D_s: 5 x 5 -> 2 x 2, D_t: 5 x 5 -> 2 x 2 -> 1 x 1) and 72 x 72 (D_s: 9 x 9 -> 4 x 4): maps that become odd inside the
    towers are floored by the pooling like F.avg_pool2d; outputs and every parameter / input gradient against the oracle."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    torch.manual_seed(size)
    B, T, k, ncls, ch = 2, 8, 3, 3, 2
    for kind in ("s", "t"):
        net = (SpatialDiscriminator if kind == "s" else TemporalDiscriminator)(ch, ncls, compute_dtype=torch.float32).to(DEV).train()
        with torch.no_grad():
            (net.attn if kind == "s" else net.self_attn).gamma.fill_(0.7)
        sd = O.make_state({kk: v.detach().cpu().clone() for kk, v in net.state_dict().items()})
        x = (torch.rand(B, k, 3, size, size) if kind == "s" else torch.rand(B, 3, T, size // 2, size // 2)) * 2 - 1
        cls = torch.randint(0, ncls, (B,))
        xo = x.clone().requires_grad_(True)
        want = (O.spatial_disc if kind == "s" else O.temporal_disc)(sd, xo, cls)
        gy = torch.randn_like(want)
        want.backward(gy)
        xg = x.to(DEV).requires_grad_(True)
        got = net(xg, cls.to(DEV))
        got.backward(gy.to(DEV))
        assert rel_l2(got, want.detach()) < 1e-4, (kind, size)
        assert rel_l2(xg.grad, xo.grad) < 2e-4, (kind, size)
        for kk, p in net.named_parameters():
            if p.grad is not None:
                v = sd[kk].grad
                assert float((p.grad.double().cpu() - v.double()).norm()) < 2e-4 * float(v.norm()) + 1e-7, (kind, size, kk)


def test_bf16_step_is_bitwise_reproducible():
    """Two Trainers built from one seed take the same two bf16 steps (same clips, same RNG draws) at the benchmark's widths
    (ch=32: the MFMA attention, the filter-row weight-gradient kernels, the ConvGRU wavefront with its in-launch split-K combine):
    the six losses, every gradient the optimizers see and every parameter / spectral-norm / batch-norm buffer afterwards are
    BIT-EQUAL.  Round 5: every weight, bias, condition and gamma gradient is summed in a fixed order (slice workspaces, per-block
    partials, first-occurrence gathers) -- no fp32 atomics with more than one contribution per address remain on this path
    (tools/repro_probe.py lists what differs; the fp64 batch-statistics atomics are the one exception)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("repro_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "tools", "repro_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    a = probe.run(32, 8, 2, 7, 2)
    b = probe.run(32, 8, 2, 7, 2)
    assert a[0] == b[0], (a[0], b[0])
    assert len(a[1]) > 600
    bad = [k for k in a[1] if not torch.equal(a[1][k], b[1][k])]
    assert not bad, bad[:10]
