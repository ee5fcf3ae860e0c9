import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    """-> dict of numpy arrays (flat keys, see tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(store, prefix):
    """All entries under `prefix.` with the prefix stripped."""
    p = prefix + "."
    return {k[len(p):]: v for k, v in store.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def from_bf16_bits(a):
    """int16 array holding the upper halves of float32 words -> float32 array of the same shape."""
    return (a.astype(np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def latent_dim_of(g):
    """Generator latent_dim of a trainer fixture (frames are 16 * latent_dim pixels; 4 unless the fixture says otherwise)."""
    return int(g["meta.latent_dim"]) if "meta.latent_dim" in g else 4


def fixture_hidden(g):
    """Initial ConvGRU states of a `meta.hidden` fixture (F15): per ConvGRU the list of per-layer numpy arrays
    [B, hidden_l, S, S] -- the closed forms `hidden.<gru>.<layer>` of synth.py, scale 0.5, as make_golden.py installed them."""
    sys.path.insert(0, GOLDEN)
    import synth
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    ld = latent_dim_of(g)
    out = []
    for gi in range(4):
        c = (8 * ch, 16 * ch, 8 * ch) if gi < 3 else (4 * ch, 8 * ch, 4 * ch)
        S = ld << gi
        out.append([synth.uniform(f"hidden.{gi}.{l}", (B, c[l], S, S), 0.5) for l in range(3)])
    return out


def full_states(g):
    """Initial state_dicts (numpy) of G / D_s / D_t for a trainer fixture: stored entries verbatim, large tensors of a
    `meta.synth` fixture from the closed forms of tests/golden/synth.py (the generating script installed the same values
    into the reference)."""
    sys.path.insert(0, GOLDEN)
    import synth
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    from dvd_gan_amd.gen_net import Generator
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    if any(kk.startswith("G.sd0b.") for kk in g):            # bf16-representable state stored in 2 bytes per value (F14)
        return [{**sub(g, tag + ".sd0"), **{kk: from_bf16_bits(v) for kk, v in sub(g, tag + ".sd0b").items()}}
                for tag in ("G", "Ds", "Dt")]
    if not int(g.get("meta.synth", 0)):
        return [sub(g, tag + ".sd0") for tag in ("G", "Ds", "Dt")]
    import torch
    with torch.device("meta"):                       # shapes only: no initialisation work
        nets = (Generator(z_dim, latent_dim_of(g), n_class, ch, T), SpatialDiscriminator(ch, n_class),
                TemporalDiscriminator(ch, n_class))
    out = []
    for net, tag in zip(nets, ("G", "Ds", "Dt")):
        tmpl = {kk: tuple(v.shape) for kk, v in net.state_dict().items()}
        out.append(synth.fill_state(g, tmpl, tag))
    return out


def fixture_real(g, i):
    """Input clips [B,3,T,S,S] of batch i of a trainer fixture (stored, or the closed form for `meta.synth` fixtures)."""
    if f"in.real.{i}" in g:
        return g[f"in.real.{i}"]
    if f"in.realb.{i}" in g:
        return from_bf16_bits(g[f"in.realb.{i}"])
    sys.path.insert(0, GOLDEN)
    import synth
    ch, T, k, B, n_class, steps, z_dim = [int(v) for v in g["meta.cfg"]]
    fr = 16 * latent_dim_of(g)
    return synth.uniform(f"real.{i}", (B, 3, T, fr, fr))
