"""CPU checks of the C-ABI library: it loads, exports every symbol include/dvdgan_hip.h declares, reports
its ABI version, validates arguments without touching a GPU, and the Python host layer reproduces the
reference state_dict layout.  (No compute calls: there is no GPU in this tier.)"""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dvdgan_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dvd_gan_amd import lib as L
    lib = L.lib()
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.dvd_abi_version() == L.ABI_VERSION
    assert b"unsupported shape" in lib.dvd_strerror(-2)


def test_argument_validation_needs_no_gpu():
    from dvd_gan_amd import lib as L
    lib = L.lib()
    d = L.ConvDesc()                        # all-zero descriptor: null pointers
    assert lib.dvd_conv_forward(ctypes.byref(d), None) == -1
    w = L.WgradDesc()
    assert lib.dvd_conv_wgrad(ctypes.byref(w), None) == -1
    assert lib.dvd_conv_pick_nsplit(L.BF16, ctypes.c_longlong(1024), 512, 512, 25) == 8       # capped
    assert lib.dvd_conv_pick_nsplit(L.BF16, ctypes.c_longlong(65536), 512, 256, 25) == 1
    with pytest.raises(RuntimeError):
        L.check(-2)


def test_missing_library_fails_loudly(monkeypatch):
    from dvd_gan_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libdvdgan_hip.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        L.lib()


def test_state_dict_layout_matches_reference(golden):
    """Key names and shapes of the three networks == the reference's (golden sd0 of F9)."""
    from conftest import sub
    from dvd_gan_amd.gen_net import Generator
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    g = golden("f9_trainer_hinge")
    nets = {"G": Generator(16, 4, 3, 2, 8), "Ds": SpatialDiscriminator(2, 3), "Dt": TemporalDiscriminator(2, 3)}
    for tag, net in nets.items():
        want = sub(g, tag + ".sd0")
        have = net.state_dict()
        assert set(have) == set(want), (tag, set(have) ^ set(want))
        for k, v in want.items():
            assert tuple(have[k].shape) == tuple(v.shape), (tag, k)
        # u / v are parameters that do not train (Normalization.py:49-50)
        for k, p in net.named_parameters():
            assert p.requires_grad == (not k.endswith(("weight_u", "weight_v"))), k


def test_condition_index_table_reproduces_quirk1():
    """samp[t*B+b] must be the row the reference's condition.repeat(T,1) gives frame b*T+t."""
    B, T = 3, 4
    t_idx = torch.arange(T).view(T, 1)
    b_idx = torch.arange(B).view(1, B)
    samp = ((b_idx * T + t_idx) % B).reshape(-1)
    cond_rows = torch.arange(B).repeat(T)           # condition.repeat(T,1): row i holds sample i % B
    for t in range(T):
        for b in range(B):
            assert int(samp[t * B + b]) == int(cond_rows[b * T + t])
