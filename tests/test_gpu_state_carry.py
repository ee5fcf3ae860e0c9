"""GPU parity of BASELINE configs[4] as a STEP: `Trainer.train_step(hidden=...)` -- initial ConvGRU states supplied at the first
frame of every ConvGRU of the generator (ConvGRU.py:104-118), T=12 (D_t pools to T'=3), 128x128 frames (latent_dim 8) --
against golden F15 = two steps of the reference Trainer driven the same way (tests/golden/make_golden.py f15).

  exact mode (f32 MFMA): six losses (2e-3 relative at step 0), |grad| checksums of every parameter of the three networks at
      their step-0 updates (1e-2), the named gradients and the gradients wrt the twelve supplied states (rel-L2 1e-2), SN u / v
      and BN statistics after two steps.
  bf16 mode (the timed mode): the same two steps at the REFERENCE's learning rate (5e-5; F15 was recorded at 40x that so that
      the exact mode's second step is a sensitive check -- at 2e-3 Adam's sign-like first update turns bf16 noise on near-zero
      gradient elements into whole +-lr weight changes) against the CPU oracle run on the same state / draws at that rate --
      the oracle whose lr-2e-3 run is pinned on F15 by tests/test_oracle_golden.py: six losses within 2e-2 absolute at both
      steps, discriminator gradients, named generator gradients and the twelve state gradients by cosine.
Measured values go to gpurun_out/state_carry_numbers.json when that directory exists.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, fixture_hidden, fixture_real, full_states, latent_dim_of, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"
NUMBERS = {}


def _dump():
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "state_carry_numbers.json"), "w") as f:
            json.dump(NUMBERS, f, indent=1, sort_keys=True)


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def make_trainer(g, dtype, lr=None):
    from dvd_gan_amd.train_step import Trainer
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    lr = float(g["meta.lr"]) if lr is None else lr
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=z_dim, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=lr, d_lr=lr, beta1=0.0, beta2=0.9,
                             n_class=n_class, k_sample=k)
    tr = Trainer([], cfg, device=torch.device(DEV), compute_dtype=dtype, latent_dim=latent_dim_of(g))
    for net, sd in zip((tr.G, tr.D_s, tr.D_t), full_states(g)):
        net.load_state_dict({kk: torch.as_tensor(v) for kk, v in sd.items()})
        net.train()
    return tr, steps


def run(g, dtype, lr=None):
    """-> per step: losses, parameter gradients at each optimizer step, gradients of the supplied states"""
    tr, steps = make_trainer(g, dtype, lr)
    hidden = [[torch.as_tensor(h).to(DEV).requires_grad_(True) for h in hs] for hs in fixture_hidden(g)]
    snaps = {}
    for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
        def stepper(net=net, tag=tag, orig=opt.step):
            snaps[tag] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            orig()
        opt.step = stepper
    out = []
    for s in range(steps):
        draws = {"perm_real": g[f"in.perm_real.{s}"], "z": g[f"in.z.{s}"], "z_class": g[f"in.z_class.{s}"],
                 "perm_fake": g[f"in.perm_fake.{s}"]}
        losses = tr.train_step(torch.as_tensor(fixture_real(g, s)), torch.as_tensor(g[f"in.labels.{s}"]), draws, hidden=hidden)
        hg = [[h.grad.detach().clone() for h in hs] for hs in hidden]
        for hs in hidden:
            for h in hs:
                h.grad = None
        out.append(([float(v.detach()) for v in losses], {t: dict(v) for t, v in snaps.items()}, hg))
    return tr, out


def test_state_carry_step_exact_matches_reference(golden):
    g = golden("f15_state_carry")
    assert int(g["meta.hidden"]) == 1 and int(g["meta.cfg"][1]) == 12 and latent_dim_of(g) == 8
    tr, out = run(g, torch.float32)
    for s, (losses, snaps, hg) in enumerate(out):
        want = g[f"out.losses.{s}"]
        NUMBERS[f"exact.losses.{s}"] = [losses, [float(v) for v in want]]
        # step 0: a pure forward / backward comparison; step 1 has been through one Adam update at lr 2e-3, whose sign-like
        # first step amplifies rounding differences (same bounds as tests/test_gpu_trainer.py::run)
        if s == 0:
            np.testing.assert_allclose(losses, want, rtol=2e-3, atol=2e-4, err_msg="losses step 0")
        else:
            np.testing.assert_allclose(losses, want, rtol=1e-2, atol=2e-2, err_msg=f"losses step {s}")
    losses, snaps, hg = out[0]
    worst = {}
    for tag in ("Ds", "Dt", "G"):
        keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
        ref = g[f"out.gsum.0.{tag}"]
        got = np.array([float(snaps[tag][kk].double().abs().sum()) for kk in keys])
        big = ref > 1e-3 * ref.max()
        err = np.abs(got[big] - ref[big]) / ref[big]
        worst[tag] = float(err.max())
        assert err.max() < 1e-2, (tag, keys[int(np.argmax(err))], float(err.max()))
        for kk, v in sub(g, f"grad.0.{tag}").items():
            r = rel(snaps[tag][kk].reshape(-1)[:v.size], v)
            worst[tag + "." + kk] = r
            assert r < 1e-2, (tag, kk, r)
    n = 0
    sums = []
    for gi, hs in enumerate(hg):
        for l, h in enumerate(hs):
            v = g[f"hgrad.0.{gi}.{l}"]
            r = rel(h.reshape(-1)[:v.size], v)
            worst[f"dh0.{gi}.{l}"] = r
            assert r < 1e-2, ("d/dh0", gi, l, r)
            sums.append(float(h.double().abs().sum()))
            n += 1
    assert n == 12
    np.testing.assert_allclose(sums, g["out.hgsum.0"], rtol=1e-2)
    NUMBERS["exact.worst"] = worst
    for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):          # SN u / v and BN statistics after two steps
        sd = net.state_dict()
        for kk, v in sub(g, tag + ".sd1").items():
            if not kk.endswith("num_batches_tracked"):
                assert rel(sd[kk], v) < 5e-2, (tag, kk)
    _dump()


def oracle_steps(g, lr):
    """The CPU oracle on F15's state, clips, draws and supplied states at learning rate `lr`: per step (losses, gradients at the
    three optimizer steps, state gradients)."""
    from oracle import dvdgan_cpu as O
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sds = full_states(g)
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, latent_dim=latent_dim_of(g), adv="hinge", g_lr=lr, d_lr=lr)
    hidden = [[torch.as_tensor(h).requires_grad_(True) for h in hs] for hs in fixture_hidden(g)]
    snaps = O.snapshot_grads(st)
    out = []
    t = torch.as_tensor
    for s in range(steps):
        losses = O.train_step(st, t(fixture_real(g, s)), t(g[f"in.labels.{s}"]), t(g[f"in.z.{s}"]), t(g[f"in.z_class.{s}"]),
                              g[f"in.perm_real.{s}"], g[f"in.perm_fake.{s}"], hidden=hidden)
        hg = [[h.grad.detach().clone() for h in hs] for hs in hidden]
        for hs in hidden:
            for h in hs:
                h.grad = None
        out.append((losses, {tg: {kk: v.clone() for kk, v in d.items()} for tg, d in snaps.items()}, hg))
    return out


def test_state_carry_step_bf16(golden):
    g = golden("f15_state_carry")
    lr = 5e-5
    want = oracle_steps(g, lr)
    np.testing.assert_allclose(want[0][0][:4], g["out.losses.0"][:4], rtol=2e-4, atol=2e-5)   # (the D losses of step 0 do not depend on lr)
    tr, out = run(g, torch.bfloat16, lr)
    num = {}
    for s, ((losses, snaps, hg), (wl, wsn, whg)) in enumerate(zip(out, want)):
        NUMBERS[f"bf16.losses.{s}"] = [losses, [float(v) for v in wl]]
        num[f"loss_err.{s}"] = float(np.abs(np.array(losses) - np.array(wl)).max())
        for tag in ("Ds", "Dt", "G"):
            a = torch.cat([snaps[tag][kk].reshape(-1).double().cpu() for kk in sorted(wsn[tag])])
            b = torch.cat([wsn[tag][kk].reshape(-1).double() for kk in sorted(wsn[tag])])
            num[f"cos.{s}.{tag}"] = cosine(a, b)
        for gi, hs in enumerate(hg):
            for l, h in enumerate(hs):
                num[f"cos.{s}.dh0.{gi}.{l}"] = cosine(h, whg[gi][l])
        for kk in sorted(sub(g, "grad.0.G")):
            num[f"cos.{s}.G.{kk}"] = cosine(snaps["G"][kk], wsn["G"][kk])
    NUMBERS["bf16"] = num
    _dump()
    # measured (ch=2: 8..32-channel layers, weights and states NOT bf16-representable; step 1 sits behind an Adam update).  Since
    # round 5 the step is bit-reproducible from run to run (every gradient is summed in a fixed order, see
    # test_bf16_step_is_bitwise_reproducible), so these are single numbers, not ranges: losses 2.6e-3 / 3.7e-3; discriminator
    # gradients 0.99986 / 0.99991 at step 0 and 0.99997 / 0.99856 at step 1; generator gradient as one vector 0.9992 / 0.99985;
    # the gradients that travel back through the recurrences -- the twelve state gradients and the early layers' weights --
    # 0.962 ... 0.992, the last stage's weights >= 0.9994 (cf. DESIGN.md section 2 / profiles/HISTORY.md on the one-ulp sensitivity of the early layers).
    # (Round 4 needed a 0.98 sanity bound at step 1: D_t scattered 0.9941 ... 0.9986 over ten runs through fp32 atomics.)
    for s in range(len(out)):
        assert num[f"loss_err.{s}"] <= 2e-2, (s, num[f"loss_err.{s}"])
        dbound = 0.999 if s == 0 else 0.997
        assert num[f"cos.{s}.Ds"] >= dbound and num[f"cos.{s}.Dt"] >= dbound, (s, num[f"cos.{s}.Ds"], num[f"cos.{s}.Dt"])
        assert num[f"cos.{s}.G"] >= (0.995 if s == 0 else 0.998), (s, num[f"cos.{s}.G"])
    for key, c in num.items():
        if not key.startswith("cos.0."):
            continue
        kk = key.split(".", 2)[2]
        if kk.startswith(("G.conv.9", "G.conv.10", "G.conv.11", "G.colorize")):
            assert c >= 0.998, (key, c)
        elif kk.startswith(("dh0.", "G.")):
            assert c >= 0.9, (key, c)
