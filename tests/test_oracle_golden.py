"""Pins the CPU oracle (oracle/dvdgan_cpu.py) against vectors produced by the real reference
(tests/golden/make_golden.py).  Tolerance: rel <= 1e-5 (same op order => usually bitwise)."""
import numpy as np
import pytest
import torch

from conftest import fixture_hidden, fixture_real, full_states, latent_dim_of, sub
from oracle import dvdgan_cpu as O

RTOL, ATOL = 1e-5, 1e-6


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def t(a):
    return torch.as_tensor(a)


def check_grads(sd, grads, rtol=1e-4, atol=1e-6):
    n = 0
    for k, g in grads.items():
        if k in sd and sd[k].grad is not None:
            close(sd[k].grad, g, rtol, atol, what="grad " + k)
            n += 1
    assert n > 0


def check_state(sd, want, rtol=RTOL):
    for k, v in want.items():
        close(sd[k], v, rtol, 1e-6, what="state " + k)


# ------------------------------------------------------------------ F1
@pytest.mark.parametrize("case", ["conv2d", "conv3d", "linear", "embed"])
def test_f1_spectral_norm(golden, case):
    g = sub(golden("f1_spectral_norm"), case)
    sd = O.make_state({"module." + k[len("module."):]: v for k, v in sub(g, "sd0").items()})
    w = O.sn_weight(sd, "module.")
    close(w, g["out.w1"], what="W/sigma after 1 forward")
    check_state(sd, {k: v for k, v in sub(g, "sd1").items() if k.endswith(("_u", "_v"))})
    x, gy = t(g["in.x"]), t(g["in.gy"])
    if case == "conv2d":
        y = torch.nn.functional.conv2d(x, w, sd["module.bias"], padding=1)
    elif case == "conv3d":
        y = torch.nn.functional.conv3d(x, w, sd["module.bias"], padding=1)
    elif case == "linear":
        y = torch.nn.functional.linear(x, w, sd["module.bias"])
    else:
        y = torch.nn.functional.embedding(x, w)
    close(y, g["out.y1"])
    y.backward(gy)
    check_grads(sd, sub(g, "grad"))
    O.sn_weight(sd, "module.")
    O.sn_weight(sd, "module.")
    check_state(sd, {k: v for k, v in sub(g, "sd3").items() if k.endswith(("_u", "_v"))})


# ------------------------------------------------------------------ F2
def test_f2_conditional_norm(golden):
    g = golden("f2_conditional_norm")
    sd = O.make_state(sub(g, "sd0"))
    x, c = t(g["in.x"]).requires_grad_(True), t(g["in.cond"]).requires_grad_(True)
    y = O.conditional_norm(sd, "", x, c, training=True)
    close(y, g["out.y_train"])
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-4)
    close(c.grad, g["grad.cond"], 1e-4)
    check_grads(sd, sub(g, "grad"))
    check_state(sd, sub(g, "sd1"))
    close(O.conditional_norm(sd, "", x, c, training=False), g["out.y_eval"])


# ------------------------------------------------------------------ F3
@pytest.mark.parametrize("tag,up", [("up1", 1), ("up2", 2)])
def test_f3_gresblock(golden, tag, up):
    g = sub(golden("f3_gresblock"), tag)
    sd = O.make_state(sub(g, "sd0"))
    x, c = t(g["in.x"]).requires_grad_(True), t(g["in.cond"]).requires_grad_(True)
    y = O.gresblock(sd, "", x, c, up)
    close(y, g["out.y"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-4, 1e-5)
    close(c.grad, g["grad.cond"], 1e-4, 1e-5)
    check_grads(sd, sub(g, "grad"), 1e-4, 1e-5)
    check_state(sd, sub(g, "sd1"))


# ------------------------------------------------------------------ F4
def test_f4_convgru_cell(golden):
    g = sub(golden("f4_convgru"), "cell")
    sd = O.make_state(sub(g, "sd0"))
    x, h = t(g["in.x"]).requires_grad_(True), t(g["in.h"]).requires_grad_(True)
    close(O.convgru_cell(sd, "", x, None), g["out.y_h0"], 1e-4, 1e-5)
    y = O.convgru_cell(sd, "", x, h)
    close(y, g["out.y"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-4, 1e-5)
    close(h.grad, g["grad.h"], 1e-4, 1e-5)
    check_grads(sd, sub(g, "grad"), 1e-4, 1e-5)


def test_f4_convgru_stack(golden):
    g = sub(golden("f4_convgru"), "gru")
    sd = O.make_state(sub(g, "sd0"))
    xs = t(g["in.xs"]).requires_grad_(True)
    hidden, outs = None, []
    for i in range(xs.shape[0]):
        hidden = O.convgru(sd, "", xs[i], hidden)
        outs.append(hidden[-1])
    y = torch.stack(outs)
    close(y, g["out.y"], 1e-4, 1e-5)
    for l in range(3):
        close(hidden[l], g[f"out.h_last.{l}"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(xs.grad, g["grad.xs"], 1e-4, 1e-5)
    check_grads(sd, sub(g, "grad"), 1e-4, 1e-5)


# ------------------------------------------------------------------ F5
@pytest.mark.parametrize("tag", ["n16", "n64"])
def test_f5_attention2d(golden, tag):
    g = sub(golden("f5_attention"), tag)
    sd = O.make_state(sub(g, "sd0"))
    x = t(g["in.x"]).requires_grad_(True)
    y = O.self_attention_2d(sd, "", x)
    close(y, g["out.y"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-4, 1e-5)
    check_grads(sd, sub(g, "grad"), 1e-4, 1e-5)


def test_f5_attention3d(golden):
    g = sub(golden("f5_attention"), "attn3d")
    sd = O.make_state(sub(g, "sd0"))
    x = t(g["in.x"]).requires_grad_(True)
    y = O.self_attention_3d(sd, "", x)
    close(y, g["out.y"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-4, 1e-5)


def test_f5_separable_attention(golden):
    """SeparableAttn (Attention.py:8-111), reference fixture: output, input gradient, every parameter gradient."""
    g = sub(golden("f5_attention"), "sep")
    sd = O.make_state(sub(g, "sd0"))
    x = t(g["in.x"]).requires_grad_(True)
    y = O.separable_attn(sd, "", x)
    close(y, g["out.y"], 1e-5, 1e-6)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-4, 1e-6)
    check_grads(sd, {k: v for k, v in sub(g, "grad").items() if k != "x"}, 1e-4, 1e-6)


# ------------------------------------------------------------------ F6
@pytest.mark.parametrize("tag,ld,T", [("a", 2, 4), ("b", 4, 4)])
def test_f6_generator(golden, tag, ld, T):
    g = sub(golden("f6_generator"), tag)
    sd = O.make_state(sub(g, "sd0"))
    z, cls = t(g["in.z"]), t(g["in.cls"])
    y = O.generator(sd, z, cls, ch=2, n_frames=T, latent_dim=ld)
    close(y, g["out.y"], 1e-4, 2e-5)
    y.backward(t(g["in.gy"]))
    check_grads(sd, sub(g, "grad"), 2e-3, 2e-5)
    check_state(sd, sub(g, "sd1"), 1e-4)
    with torch.no_grad():
        close(O.generator(sd, z, cls, ch=2, n_frames=T, latent_dim=ld, training=False),
              g["out.y_eval"], 1e-4, 2e-5)
    check_state(sd, sub(g, "sd2"), 1e-4)     # SN advanced in eval too (quirk 2)


# ------------------------------------------------------------------ F7
def test_f7_spatial(golden):
    G = golden("f7_discriminators")
    g = sub(G, "ds")
    sd = O.make_state(sub(g, "sd0"))
    x, cls = t(g["in.x"]).requires_grad_(True), t(g["in.cls"])
    y = O.spatial_disc(sd, x, cls)
    close(y, g["out.y"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-3, 1e-6)
    check_grads(sd, sub(g, "grad"), 1e-3, 1e-5)
    check_state(sd, {k: v for k, v in sub(g, "sd1").items() if k.endswith(("_u", "_v"))})
    with torch.no_grad():
        close(O.spatial_disc(sd, t(G["ds32.in.x"]), cls), G["ds32.out.y"], 1e-4, 1e-5)


def test_f7_temporal(golden):
    G = golden("f7_discriminators")
    g = sub(G, "dt")
    sd = O.make_state(sub(g, "sd0"))
    x, cls = t(g["in.x"]).requires_grad_(True), t(g["in.cls"])
    y = O.temporal_disc(sd, x, cls)
    close(y, g["out.y"], 1e-4, 1e-5)
    y.backward(t(g["in.gy"]))
    close(x.grad, g["grad.x"], 1e-3, 1e-6)
    check_grads(sd, sub(g, "grad"), 1e-3, 1e-5)
    assert int(G["dt16.meta.raises"]) == 1
    with pytest.raises(RuntimeError):      # quirk 4: D_t needs frames >= 64x64 (32x32 after downsample)
        O.temporal_disc(sd, torch.rand(1, 3, 8, 16, 16), cls[:1])


# ------------------------------------------------------------------ F8
def test_f8_helpers(golden):
    g = golden("f8_helpers")
    data = t(g["in.data"])
    close(O.sample_k_frames(data, O.frame_ids_from_perm(g["in.perm"], 4)), g["out.sample_k4"], 0, 0)
    close(O.sample_k_frames(data, O.frame_ids_from_perm(g["in.perm_k9"], 9)), g["out.sample_k9"], 0, 0)
    close(O.vid_downsample(data), g["out.down"], 1e-6, 1e-7)


# ------------------------------------------------------------------ F9 / F10
def _run_steps(g, inputs, adv, check_named=True):
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in inputs["meta.cfg"]] if "meta.cfg" in inputs \
        else [int(v) for v in g["meta.cfg"]]
    lr = float(g["meta.lr"])
    sds = full_states(inputs)
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, adv=adv, g_lr=lr, d_lr=lr)
    nb = len([k_ for k_ in inputs if k_.startswith("in.labels.")])
    for s in range(steps):
        losses = O.train_step(st, t(fixture_real(inputs, s % nb)), t(inputs[f"in.labels.{s % nb}"]),
                              t(inputs[f"in.z.{s}"]), t(inputs[f"in.z_class.{s}"]),
                              inputs[f"in.perm_real.{s}"], inputs[f"in.perm_fake.{s}"])
        np.testing.assert_allclose(losses, g[f"out.losses.{s}"], rtol=2e-4, atol=2e-5,
                                   err_msg=f"losses step {s}")
        # post-step parameter checksums of every tensor (sum |p|), recorded right after each
        # optimizer.step(); G's is taken last so all three can be compared at step end only for G.
        keys = [str(x) for x in g["meta.psum_keys.G"]]
        got = np.array([float(st.G[kk].double().abs().sum()) for kk in keys])
        np.testing.assert_allclose(got, g[f"out.psum.{s}.G"], rtol=5e-5, err_msg=f"G psum step {s}")
    for tag, sd in (("G", st.G), ("Ds", st.Ds), ("Dt", st.Dt)):
        check_state(sd, sub(g, tag + ".sd1"), 1e-3)
    return st


def test_f9_trainer_hinge(golden):
    g = golden("f9_trainer_hinge")
    st = _run_steps(g, g, "hinge")
    # named gradients of the last step's G update
    for k, v in sub(g, "grad.1.G").items():
        close(st.G[k].grad, v, 2e-3, 1e-6, what="G grad " + k)


def test_f9_trainer_wgangp(golden):
    g, base = golden("f9_trainer_wgangp"), golden("f9_trainer_hinge")
    _run_steps(g, base, "wgan-gp")


def test_f10_config1_plumbing(golden):
    g = golden("f10_config1")
    _run_steps(g, g, "hinge")


def test_f13_two_discriminator_iterations(golden):
    """trainer.py:230 with d_iters = 2, run by the real reference: the oracle's loop consumes the recorded draws of both
    iterations; losses (last iteration + generator), generator checksums, named gradients and the final states are compared."""
    g = golden("f13_two_d_iters")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    D = int(g["meta.d_iters"])
    assert D == 2
    lr = float(g["meta.lr"])
    sds = full_states(g)
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, adv="hinge", g_lr=lr, d_lr=lr)
    for s in range(steps):
        losses = O.train_step(st, t(fixture_real(g, s)), t(g[f"in.labels.{s}"]),
                              [t(g[f"in.z.{s}.{i}"]) for i in range(D)], [t(g[f"in.z_class.{s}.{i}"]) for i in range(D)],
                              [g[f"in.perm_real.{s}.{i}"] for i in range(D)], [g[f"in.perm_fake.{s}.{i}"] for i in range(D)],
                              d_iters=D)
        np.testing.assert_allclose(losses, g[f"out.losses.{s}"], rtol=2e-4, atol=2e-5, err_msg=f"losses step {s}")
        keys = [str(x) for x in g["meta.psum_keys.G"]]
        got = np.array([float(st.G[kk].double().abs().sum()) for kk in keys])
        np.testing.assert_allclose(got, g[f"out.psum.{s}.G"], rtol=5e-5, err_msg=f"G psum step {s}")
    for kk, v in sub(g, "grad.1.G").items():
        close(st.G[kk].grad, v, 2e-3, 1e-6, what="G grad " + kk)
    for tag, sd in (("G", st.G), ("Ds", st.Ds), ("Dt", st.Dt)):
        check_state(sd, sub(g, tag + ".sd1"), 1e-3)


# ------------------------------------------------------------------ F11: the benchmark's real channel widths
def test_f11_full_width_step(golden):
    """The oracle against ONE step of the real reference at ch=32, T=48, 64x64, 101 classes, k=8, B=2 (BASELINE
    configs[1] per-clip shape): six losses, |grad| checksum of every parameter of the three networks at its optimizer
    step, and the stored heads of the named gradients."""
    g = golden("f11_full_width")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    sds = full_states(g)
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, adv="hinge", g_lr=float(g["meta.lr"]), d_lr=float(g["meta.lr"]))
    snaps = O.snapshot_grads(st)
    losses = O.train_step(st, t(fixture_real(g, 0)), t(g["in.labels.0"]), t(g["in.z.0"]), t(g["in.z_class.0"]),
                          g["in.perm_real.0"], g["in.perm_fake.0"])
    np.testing.assert_allclose(losses, g["out.losses.0"], rtol=2e-4, atol=2e-5)
    for tag in ("Ds", "Dt", "G"):
        keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
        got = np.array([float(snaps[tag][kk].double().abs().sum()) for kk in keys])
        np.testing.assert_allclose(got, g[f"out.gsum.0.{tag}"], rtol=2e-3, err_msg=f"{tag} gradient checksums")
        for kk, v in sub(g, f"grad.0.{tag}").items():
            close(snaps[tag][kk].reshape(-1)[:v.size], v, 2e-3, 1e-7, what=f"{tag} grad {kk}")


# ------------------------------------------------------------------ F14: the reference's own default initialisation
def test_f14_default_init_step(golden):
    """The oracle against ONE step of the real reference started from ITS OWN default initialisation (orthogonal ConvGRU
    weights, ConvGRU.py:20-26) at ch=4, T=16, B=2 -- every weight and clip bf16-representable, stored in 2 bytes: losses,
    |grad| checksums of every parameter, heads of the named gradients, post-step SN / BN state."""
    g = golden("f14_default_init_bf16")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    sds = full_states(g)
    for sd in sds:                                   # the decoder restores exactly bf16-representable float32 values
        for kk, v in sd.items():
            if v.dtype == np.float32:
                assert np.array_equal(t(v).to(torch.bfloat16).float().numpy(), v), kk
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, adv="hinge", g_lr=float(g["meta.lr"]), d_lr=float(g["meta.lr"]))
    snaps = O.snapshot_grads(st)
    losses = O.train_step(st, t(fixture_real(g, 0)), t(g["in.labels.0"]), t(g["in.z.0"]), t(g["in.z_class.0"]),
                          g["in.perm_real.0"], g["in.perm_fake.0"])
    np.testing.assert_allclose(losses, g["out.losses.0"], rtol=2e-4, atol=2e-5)
    for tag in ("Ds", "Dt", "G"):
        keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
        got = np.array([float(snaps[tag][kk].double().abs().sum()) for kk in keys])
        np.testing.assert_allclose(got, g[f"out.gsum.0.{tag}"], rtol=2e-3, atol=1e-9, err_msg=f"{tag} gradient checksums")
        for kk, v in sub(g, f"grad.0.{tag}").items():
            close(snaps[tag][kk].reshape(-1)[:v.size], v, 2e-3, 1e-7, what=f"{tag} grad {kk}")
    for tag, sd in (("G", st.G), ("Ds", st.Ds), ("Dt", st.Dt)):
        check_state(sd, sub(g, tag + ".sd1"), 1e-3)


# ------------------------------------------------------------------ F15: state carry as a full step (BASELINE configs[4])
def test_f15_state_carry_steps(golden):
    """The oracle's `train_step(hidden=)` against two steps of the reference Trainer whose generator received initial ConvGRU
    states at the first frame of every ConvGRU (ConvGRU.py:104-118): T=12, 128x128 frames, ch=2, k=4, B=2.  Losses, the
    |grad| checksums of every parameter, the named gradients, the gradients wrt the twelve supplied states and the post-step
    SN / BN state."""
    g = golden("f15_state_carry")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    assert int(g["meta.hidden"]) == 1 and latent_dim_of(g) == 8 and T == 12
    lr = float(g["meta.lr"])
    sds = full_states(g)
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, latent_dim=8, adv="hinge", g_lr=lr, d_lr=lr)
    hidden = [[t(h).requires_grad_(True) for h in hs] for hs in fixture_hidden(g)]
    for s in range(steps):
        snaps = O.snapshot_grads(st) if s == 0 else snaps
        losses = O.train_step(st, t(fixture_real(g, s)), t(g[f"in.labels.{s}"]), t(g[f"in.z.{s}"]), t(g[f"in.z_class.{s}"]),
                              g[f"in.perm_real.{s}"], g[f"in.perm_fake.{s}"], hidden=hidden)
        np.testing.assert_allclose(losses, g[f"out.losses.{s}"], rtol=2e-4, atol=2e-5, err_msg=f"losses step {s}")
        for tag in ("Ds", "Dt", "G"):
            keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
            got = np.array([float(snaps[tag][kk].double().abs().sum()) for kk in keys])
            np.testing.assert_allclose(got, g[f"out.gsum.{s}.{tag}"], rtol=2e-3, atol=1e-9, err_msg=f"{tag} checksums step {s}")
            for kk, v in sub(g, f"grad.{s}.{tag}").items():
                close(snaps[tag][kk].reshape(-1)[:v.size], v, 2e-3, 1e-7, what=f"{tag} grad {kk} step {s}")
        sums = []
        for gi, hs in enumerate(hidden):
            for l, h in enumerate(hs):
                v = g[f"hgrad.{s}.{gi}.{l}"]
                close(h.grad.reshape(-1)[:v.size], v, 2e-3, 1e-8, what=f"d/dh0 gru {gi} layer {l} step {s}")
                sums.append(float(h.grad.double().abs().sum()))
                h.grad = None
        np.testing.assert_allclose(sums, g[f"out.hgsum.{s}"], rtol=2e-3)
    for tag, sd in (("G", st.G), ("Ds", st.Ds), ("Dt", st.Dt)):
        check_state(sd, sub(g, tag + ".sd1"), 1e-3)
