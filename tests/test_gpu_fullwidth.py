"""GPU parity at the REAL channel widths of the benchmark (BASELINE configs[1] per-clip shape: ch=32 -> 128..1024
channels, T=48, 64x64, 101 classes, k=8, hinge) with B=2.

  * exact (f32 MFMA) mode, one full G + D_s + D_t step against golden F11 = ONE step of the unmodified reference
    Trainer at this shape (tests/golden/make_golden.py f11): six losses, |grad| checksums of every parameter of the
    three networks, heads of 27 named gradients, SN u / v and BN statistics after the step.  This drives the dispatch the
    benchmark takes (halo / igemm / 8-wave variants, split-K recurrent convs, row weight-gradient kernels).
  * bf16 (production) mode, TEACHER-FORCED: every generator module (4 ConvGRUs over 48 steps, 8 GResBlocks, colorize)
    and both discriminators get the oracle's exact fp32 input and upstream gradient; their outputs, input gradients and
    all parameter gradients are bounded against the oracle's.  Teacher forcing isolates the error a module ADDS from the
    error it inherits (the free-running generator amplifies any perturbation ~2000x, see test_sensitivity below).
  * bf16 mode, free-running full step: the six losses and the discriminator outputs.
  * the sensitivity measurement behind the tolerances, as an asserting test.

The stated tolerances are the table in profiles/HISTORY.md (rounds 1-4, section 2; summary in DESIGN.md section 2); measured values are written to
gpurun_out/fullwidth_numbers.json when that directory exists.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, fixture_real, full_states, latent_dim_of, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"
NUMBERS = {}


def _dump():
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "fullwidth_numbers.json"), "w") as f:
            json.dump(NUMBERS, f, indent=1, sort_keys=True)


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def cfg_of(g, lr=None):
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    lr = float(g["meta.lr"]) if lr is None else lr
    return argparse.Namespace(adv_loss="hinge", z_dim=z_dim, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                              total_epoch=1, d_iters=1, batch_size=B, g_lr=lr, d_lr=lr, beta1=0.0, beta2=0.9,
                              n_class=n_class, k_sample=k)


def make_trainer(g, dtype):
    from dvd_gan_amd.train_step import Trainer
    tr = Trainer([], cfg_of(g), device=torch.device(DEV), compute_dtype=dtype, latent_dim=latent_dim_of(g))
    for net, sd in zip((tr.G, tr.D_s, tr.D_t), full_states(g)):
        net.load_state_dict({kk: torch.as_tensor(v) for kk, v in sd.items()})
        net.train()
    return tr


def draws_of(g, s=0):
    return {"perm_real": g[f"in.perm_real.{s}"], "z": g[f"in.z.{s}"], "z_class": g[f"in.z_class.{s}"],
            "perm_fake": g[f"in.perm_fake.{s}"]}


def snapshot_hip_grads(tr):
    snaps = {}
    for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
        def stepper(net=net, tag=tag, orig=opt.step):
            snaps[tag] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            orig()
        opt.step = stepper
    return snaps


# ------------------------------------------------------------------------------------------ exact mode vs the reference
@pytest.mark.parametrize("fixture", ["f11_full_width", "f16_full_width_128"])
def test_exact_step_matches_reference_at_full_width(golden, fixture):
    """f11: 48 x 64 x 64, 101 classes, B=2 (BASELINE configs[1] / [2]); f16: 48 x 128 x 128, 600 classes, B=1 (configs[3])."""
    g = golden(fixture)
    NUMBERS_ = NUMBERS.setdefault(fixture, {})
    tr = make_trainer(g, torch.float32)
    snaps = snapshot_hip_grads(tr)
    losses = [float(v.detach()) for v in tr.train_step(torch.as_tensor(fixture_real(g, 0)), torch.as_tensor(g["in.labels.0"]),
                                                      draws_of(g))]
    NUMBERS_["exact.losses"] = losses
    NUMBERS_["exact.losses_ref"] = [float(v) for v in g["out.losses.0"]]
    np.testing.assert_allclose(losses, g["out.losses.0"], rtol=2e-3, atol=2e-4)
    worst = {}
    for tag in ("Ds", "Dt", "G"):
        keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
        ref = g[f"out.gsum.0.{tag}"]
        got = np.array([float(snaps[tag][kk].double().abs().sum()) for kk in keys])
        big = ref > 1e-3 * ref.max()             # gradients that are zero in exact arithmetic are rounding noise on both sides
        err = np.abs(got[big] - ref[big]) / ref[big]
        worst[tag] = float(err.max())
        assert err.max() < 1e-2, (tag, keys[int(np.argmax(err))], float(err.max()))
        for kk, v in sub(g, f"grad.0.{tag}").items():
            r = rel(snaps[tag][kk].reshape(-1)[:v.size], v)
            worst[tag + "." + kk] = r
            if np.abs(v).max() > 1e-4 * np.abs(ref).max() / max(v.size, 1):
                assert r < 1e-2, (tag, kk, r)
    NUMBERS_["exact.grad_worst"] = worst
    for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):          # SN u / v and BN statistics after the step
        sd = net.state_dict()
        for kk, v in sub(g, tag + ".sd1").items():
            if not kk.endswith("num_batches_tracked"):
                assert rel(sd[kk], v) < 2e-3, (tag, kk)
    _dump()


# ------------------------------------------------------------------------------------------ oracle run shared by the bf16 tests
@pytest.fixture(scope="module")
def oracle_run(golden):
    """One oracle step on fixture F11 with every generator tap, discriminator output and parameter gradient kept."""
    from oracle import dvdgan_cpu as O
    g = golden("f11_full_width")
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sds = full_states(g)
    st = O.TrainState(O.make_state(sds[0]), O.make_state(sds[1]), O.make_state(sds[2]), ch=ch, n_frames=T, k_sample=k,
                      n_class=n_class, z_dim=z_dim, adv="hinge", g_lr=float(g["meta.lr"]), d_lr=float(g["meta.lr"]))
    snaps = O.snapshot_grads(st)
    rec = {}
    t = torch.as_tensor
    losses = O.train_step(st, t(fixture_real(g, 0)), t(g["in.labels.0"]), t(g["in.z.0"]), t(g["in.z_class.0"]),
                          g["in.perm_real.0"], g["in.perm_fake.0"], rec=rec)
    np.testing.assert_allclose(losses, g["out.losses.0"], rtol=2e-4, atol=2e-5)     # the checker itself agrees with the reference
    return {"g": g, "sds": sds, "rec": rec, "snaps": snaps, "losses": losses, "cfg": (ch, T, k, B, n_class, z_dim)}


def _hip_generator(o, dtype):
    from dvd_gan_amd.gen_net import Generator
    ch, T, k, B, n_class, z_dim = o["cfg"]
    G = Generator(z_dim, 4, n_class, ch, T, compute_dtype=dtype)
    G.load_state_dict({kk: torch.as_tensor(v) for kk, v in o["sds"][0].items()})
    return G.to(DEV).train()


# Stated bf16 bounds per module kind: (output rel-L2, input-gradient rel-L2, parameter-gradient cosine, parameter-gradient rel-L2)
BF16_MODULE_TOL = {"gru": (2e-2, 5e-2, 0.999, 5e-2), "res": (1e-2, 8e-2, 0.999, 5e-2), "colorize": (1e-2, 3e-2, 0.999, 5e-2)}


def _torch_bf16_gresblock_dx(o, kmod, upsample, zc_cpu):
    """Yardstick for the input gradient of a GResBlock in bf16 (it is the small difference of O(|g|) terms inside the batch
    norms' backward, so bf16 rounding of the tensors between the ops is amplified ~10x): the REFERENCE block's own ops
    (oracle.gresblock = GResBlock.py:42-86) run by plain PyTorch on the CPU with every tensor -- weights, input, condition,
    upstream gradient -- in torch.bfloat16, teacher-forced like the HIP module.  Returns rel-L2 of d/dx against the fp32 oracle."""
    from oracle import dvdgan_cpu as O
    ch, T, k, B, n_class, z_dim = o["cfg"]
    pfx = f"conv.{kmod}."
    taps = o["rec"]["taps"]
    sd = O.make_state({kk: v for kk, v in o["sds"][0].items() if kk.startswith(pfx)}, dtype=torch.bfloat16)
    x = taps[kmod].detach().to(torch.bfloat16).requires_grad_(True)
    y = O.gresblock(sd, pfx, x, zc_cpu.repeat(T, 1).to(torch.bfloat16), upsample)
    y.backward(taps[kmod + 1].grad.to(torch.bfloat16))
    return rel(x.grad.float(), taps[kmod].grad)


def test_bf16_generator_modules_teacher_forced(oracle_run):
    from dvd_gan_amd import functional as Fn
    from dvd_gan_amd import lib as L
    from dvd_gan_amd.gen_net import ConvGRU
    o = oracle_run
    ch, T, k, B, n_class, z_dim = o["cfg"]
    G = _hip_generator(o, torch.bfloat16)
    taps, gsnap = o["rec"]["taps"], o["snaps"]["G"]
    g = o["g"]
    emb = torch.as_tensor(o["sds"][0]["embedding.weight"])[torch.as_tensor(g["in.z_class.0"])]
    zc_cpu = torch.cat([torch.as_tensor(g["in.z.0"]), emb], 1)
    zc = zc_cpu.to(DEV)
    t_idx, b_idx = torch.arange(T).view(T, 1), torch.arange(B).view(1, B)
    samp = ((b_idx * T + t_idx) % B).reshape(-1).to(torch.int32).to(DEV)
    table, bad = {}, []
    for kmod, m in enumerate(G.conv):
        x_ref, y_ref = taps[kmod], taps[kmod + 1]
        is_gru = isinstance(m, ConvGRU)
        x = x_ref.detach().to(DEV).requires_grad_(True)
        if kmod == 0:
            y = m.run(Fn.ToChannelsLast.apply(x, torch.bfloat16, None), T, True)[-1]
        elif is_gru:
            y = m.run(Fn.ToChannelsLast.apply(x, torch.bfloat16, (B, T)), T, False)[-1]
        else:
            y = m.run(Fn.ToChannelsLast.apply(x, torch.bfloat16, (B, T)), zc, samp)
        y = Fn.FromChannelsLast.apply(y, y_ref.shape[1], (B, T))
        y.backward(y_ref.grad.to(DEV))
        kind = "gru" if is_gru else "res"
        t_out, t_dx, t_cos, t_rel = BF16_MODULE_TOL[kind]
        r_out, r_dx = rel(y, y_ref), rel(x.grad, x_ref.grad)
        cs, rl, worst = [], [], (1.0, "")
        scale = max(float(gsnap[f"conv.{kmod}.{kk}"].abs().max()) for kk, p in m.named_parameters() if p.grad is not None)
        for kk, p in m.named_parameters():
            if p.grad is None:
                continue
            ref = gsnap[f"conv.{kmod}.{kk}"]
            if float(ref.abs().max()) < 1e-4 * scale:       # zero in exact arithmetic (bias in front of a batch norm)
                continue
            c = cosine(p.grad, ref)
            cs.append(c); rl.append(rel(p.grad, ref))
            worst = min(worst, (c, kk))
        table[f"conv.{kmod}.{kind}"] = {"out": r_out, "dx": r_dx, "pgrad_cos_min": min(cs), "pgrad_rel_max": max(rl),
                                        "worst": worst[1]}
        ok_dx = r_dx < t_dx
        if not is_gru:
            # the input gradient of a GResBlock is judged against what plain PyTorch makes of the reference block in bf16
            # (see _torch_bf16_gresblock_dx; measured 5.1e-2 ... 8.3e-2 there against 2.8e-2 ... 5.5e-2 here): never worse than
            # that yardstick, and below the absolute cap
            yard = _torch_bf16_gresblock_dx(o, kmod, m.upsample_factor, zc_cpu)
            table[f"conv.{kmod}.{kind}"]["dx_torch_bf16"] = yard
            ok_dx = ok_dx and r_dx < yard
        NUMBERS["bf16.teacher_forced"] = table
        _dump()
        if not (r_out < t_out and ok_dx and min(cs) > t_cos and max(rl) < t_rel):
            bad.append((kmod, table[f"conv.{kmod}.{kind}"]))
        del x, y
    # colorize: relu -> SN conv 3x3 -> tanh on the last tap, output = the generated clips
    x_ref, y_ref = taps[12], o["rec"]["fake"]
    x = x_ref.detach().to(DEV).requires_grad_(True)
    y = G.colorize(Fn.ToChannelsLast.apply(x, torch.bfloat16, (B, T)), relu_in=True, act=L.ACT_TANH)
    y = Fn.FromChannelsLast.apply(y, 3, (B, T))
    y.backward(y_ref.grad.reshape(y.shape).to(DEV))
    t_out, t_dx, t_cos, t_rel = BF16_MODULE_TOL["colorize"]
    r_out, r_dx = rel(y, y_ref.reshape(y.shape)), rel(x.grad, x_ref.grad)
    c = cosine(G.colorize.module.weight_bar.grad, gsnap["colorize.module.weight_bar"])
    table["colorize"] = {"out": r_out, "dx": r_dx, "pgrad_cos_min": c}
    _dump()
    assert r_out < t_out and r_dx < t_dx and c > t_cos, table["colorize"]
    assert not bad, bad


BF16_D_TOL = {"out_abs": 2e-2, "grad_cos": 0.999}       # D outputs are O(1) logits; SURVEY section 8c: cosine >= 0.999 (D)


def test_bf16_discriminators_teacher_forced(oracle_run):
    """D_s / D_t in bf16 on the oracle's exact inputs (real clips and the oracle's generated clips): raw outputs, and the
    gradients of their own update (hinge real + fake) against the oracle's."""
    from dvd_gan_amd import functional as Fn
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    o = oracle_run
    ch, T, k, B, n_class, z_dim = o["cfg"]
    g, rec = o["g"], o["rec"]
    labels, z_class = torch.as_tensor(g["in.labels.0"]).to(DEV), torch.as_tensor(g["in.z_class.0"]).to(DEV)
    table = {}
    for tag, cls, sd, xr, xf, key in (("Ds", SpatialDiscriminator, o["sds"][1], rec["real_s"], rec["fake_s"], "ds"),
                                      ("Dt", TemporalDiscriminator, o["sds"][2], rec["real_d"], rec["fake_d"], "dt")):
        D = cls(ch, n_class, compute_dtype=torch.bfloat16)
        D.load_state_dict({kk: torch.as_tensor(v) for kk, v in sd.items()})
        D = D.to(DEV).train()
        o_r = D(xr.detach().to(DEV), labels)
        o_f = D(xf.detach().to(DEV), z_class)
        e_r = float((o_r.detach().cpu() - rec["d_out"][key + "_real"].detach()).abs().max())
        e_f = float((o_f.detach().cpu() - rec["d_out"][key + "_fake"].detach()).abs().max())
        (Fn.AdvLoss.apply(o_r, True, True) + Fn.AdvLoss.apply(o_f, False, True)).backward()
        cs = {}
        ref = o["snaps"][tag]
        scale = max(float(v.abs().max()) for v in ref.values())
        for kk, p in D.named_parameters():
            if p.grad is not None and float(ref[kk].abs().max()) > 1e-4 * scale:
                cs[kk] = cosine(p.grad, ref[kk])
        worst = min(cs, key=cs.get)
        allg = torch.cat([p.grad.reshape(-1).cpu() for kk, p in D.named_parameters() if kk in cs])
        allr = torch.cat([ref[kk].reshape(-1) for kk, p in D.named_parameters() if kk in cs])
        table[tag] = {"out_real_maxabs": e_r, "out_fake_maxabs": e_f, "out_scale": float(rec["d_out"][key + "_real"].abs().max()),
                      "grad_cos_min": cs[worst], "worst": worst, "grad_cos_all": cosine(allg, allr)}
        NUMBERS["bf16.discriminators"] = table
        _dump()
        assert e_r < BF16_D_TOL["out_abs"] and e_f < BF16_D_TOL["out_abs"], table[tag]
        assert table[tag]["grad_cos_all"] > BF16_D_TOL["grad_cos"], table[tag]
        assert cs[worst] > 0.99, table[tag]


def test_bf16_free_running_step(oracle_run):
    """The production mode end to end: one bf16 step on F11's weights / draws, nothing teacher-forced.  Stated bounds
    (SURVEY section 8c): each of the six loss terms within 1e-2 of the reference's; gradient of each network as one
    vector: cosine >= 0.999 (D_s, D_t) and >= 0.99 (G) against the oracle's; generated clips rel-L2 <= 2e-2."""
    o = oracle_run
    g = o["g"]
    tr = make_trainer(g, torch.bfloat16)
    snaps = snapshot_hip_grads(tr)
    fake = {}
    def keep(module, inputs, out):
        fake["y"] = out.detach().float().cpu()                 # (returns None: the output itself is left alone)
    hook = tr.G.register_forward_hook(keep)
    losses = [float(v.detach()) for v in tr.train_step(torch.as_tensor(fixture_real(g, 0)), torch.as_tensor(g["in.labels.0"]),
                                                      draws_of(g))]
    hook.remove()
    res = {"losses": losses, "abs_err": [abs(a - float(b)) for a, b in zip(losses, g["out.losses.0"])],
           "fake_rel": rel(fake["y"], o["rec"]["fake"])}
    for tag in ("Ds", "Dt", "G"):
        ref = o["snaps"][tag]
        keys = [kk for kk in ref if kk in snaps[tag]]
        a = torch.cat([snaps[tag][kk].reshape(-1).cpu() for kk in keys])
        b = torch.cat([ref[kk].reshape(-1) for kk in keys])
        scale = max(float(ref[kk].abs().max()) for kk in keys)
        per = {kk: cosine(snaps[tag][kk], ref[kk]) for kk in keys if float(ref[kk].abs().max()) > 1e-4 * scale}
        worst = min(per, key=per.get)
        res[tag] = {"cos_all": cosine(a, b), "rel_all": rel(a, b), "cos_min": per[worst], "worst": worst}
    NUMBERS["bf16.free_running"] = res
    _dump()
    np.testing.assert_allclose(losses, g["out.losses.0"], atol=1e-2, rtol=0)
    assert res["fake_rel"] < 2e-2, res["fake_rel"]
    assert res["Ds"]["cos_all"] > 0.999 and res["Dt"]["cos_all"] > 0.999 and res["G"]["cos_all"] > 0.99, res


def test_bf16_step_at_128x128_full_width(golden):
    """BASELINE configs[3]'s per-clip shape in the TIMED mode: one bf16 step at ch=32, T=48, 128x128, 600 classes, B=1 against the
    reference's own numbers (fixture F16; no oracle run: the reference takes four minutes per step at this size).  Bounds as for
    the 64x64 shape: six losses within 1e-2; per network, the stored heads of the named gradients as one vector
    cosine >= 0.999 (D_s, D_t) / >= 0.99 (G), and the |grad| checksums of all parameters as one vector within 2e-2 relative."""
    g = golden("f16_full_width_128")
    tr = make_trainer(g, torch.bfloat16)
    snaps = snapshot_hip_grads(tr)
    losses = [float(v.detach()) for v in tr.train_step(torch.as_tensor(fixture_real(g, 0)), torch.as_tensor(g["in.labels.0"]),
                                                      draws_of(g))]
    res = {"losses": losses, "abs_err": [abs(a - float(b)) for a, b in zip(losses, g["out.losses.0"])]}
    for tag in ("Ds", "Dt", "G"):
        named = sub(g, f"grad.0.{tag}")
        a = torch.cat([snaps[tag][kk].reshape(-1)[:v.size].double().cpu() for kk, v in named.items()])
        b = torch.cat([torch.as_tensor(v).double().reshape(-1) for v in named.values()])
        keys = [str(x) for x in g[f"meta.gsum_keys.{tag}"]]
        got = np.array([float(snaps[tag][kk].double().abs().sum()) for kk in keys])
        ref = g[f"out.gsum.0.{tag}"]
        res[tag] = {"cos_named": cosine(a, b), "gsum_rel": float(np.linalg.norm(got - ref) / np.linalg.norm(ref)),
                    "cos_each": {kk: cosine(snaps[tag][kk].reshape(-1)[:v.size], v) for kk, v in named.items()}}
    NUMBERS["bf16.f16_128x128"] = res
    _dump()
    np.testing.assert_allclose(losses, g["out.losses.0"], atol=1e-2, rtol=0)
    assert res["Ds"]["cos_named"] > 0.999 and res["Dt"]["cos_named"] > 0.999 and res["G"]["cos_named"] > 0.99, res
    for tag in ("Ds", "Dt", "G"):
        assert res[tag]["gsum_rel"] < 2e-2, (tag, res[tag]["gsum_rel"])


@pytest.mark.parametrize("init", ["fixture", "torch_default"])
def test_sensitivity_of_the_free_running_generator(oracle_run, init):
    """What bounds the generator's end-to-end OUTPUT in bf16 is the network's own conditioning, which depends on the weights:
    in exact (f32) mode a 2^-9 relative perturbation of the weights -- one bf16 ulp, the smallest error any bf16
    implementation makes -- is measured next to the bf16 mode's deviation on the same input.
      fixture        (F11 weights: uniform, bound 1/sqrt(fan_in)): both are ~1 %, the SURVEY bound of 2e-2 holds;
      torch_default  (orthogonal ConvGRU weights, ConvGRU.py:20-26, what a fresh model starts from): the 48-step recurrence
                     amplifies the same perturbation to tens of percent -- no bf16 implementation can meet 2e-2 there.
    Asserted for both: the bf16 mode deviates by no more than 3x the one-ulp weight perturbation (+5e-2 absolute), and
    mean |y| / std of the generated clips agree within 2 % between the bf16 and the exact mode."""
    from dvd_gan_amd.gen_net import Generator
    o = oracle_run
    g = o["g"]
    ch, T, k, B, n_class, z_dim = o["cfg"]
    z, zc = torch.as_tensor(g["in.z.0"]).to(DEV), torch.as_tensor(g["in.z_class.0"]).to(DEV)
    if init == "fixture":
        sd0 = {kk: torch.as_tensor(v) for kk, v in o["sds"][0].items()}
    else:
        torch.manual_seed(0)
        sd0 = {kk: v.clone() for kk, v in Generator(z_dim, 4, n_class, ch, T).state_dict().items()}
    outs = {}
    for name, dtype, eps in (("exact", torch.float32, 0.0), ("perturbed", torch.float32, 2.0 ** -9), ("bf16", torch.bfloat16, 0.0)):
        G = Generator(z_dim, 4, n_class, ch, T, compute_dtype=dtype)
        G.load_state_dict(sd0)
        G = G.to(DEV).train()
        if eps:
            gen = torch.Generator(device="cpu").manual_seed(5)
            with torch.no_grad():
                for kk, p in G.named_parameters():
                    if p.dim() > 1 and not kk.endswith(("weight_u", "weight_v")):
                        p.mul_(1 + eps * torch.randn(p.shape, generator=gen).to(DEV))
        with torch.no_grad():
            outs[name] = G(z, zc).float().cpu()
        del G
    r_pert, r_bf16 = rel(outs["perturbed"], outs["exact"]), rel(outs["bf16"], outs["exact"])
    st = {n: (float(v.abs().mean()), float(v.std())) for n, v in outs.items()}
    NUMBERS["sensitivity." + init] = {"rel_out_weights_1ulp_bf16": r_pert, "rel_out_bf16_mode": r_bf16, "stats": st}
    if init == "fixture":
        r_ref = rel(outs["exact"], o["rec"]["fake"])
        NUMBERS["sensitivity." + init]["rel_out_exact_vs_oracle"] = r_ref
        assert r_ref < 1e-3, r_ref                                    # the exact mode itself tracks the oracle
        assert r_bf16 < 2e-2, r_bf16                                  # G output rel-L2 <= 2e-2
    _dump()
    assert r_bf16 < 3 * r_pert + 0.05, (r_bf16, r_pert)
    assert abs(st["bf16"][0] - st["exact"][0]) < 0.02 * st["exact"][0] + 1e-3
    assert abs(st["bf16"][1] - st["exact"][1]) < 0.02 * st["exact"][1] + 1e-3
