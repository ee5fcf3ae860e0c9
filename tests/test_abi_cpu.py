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


def test_dispatch_queries_need_no_gpu():
    """The shape queries behind the round-4 kernels (which weight image a convolution wants, whether the MFMA attention serves a
    call, workspace sizes) are pure host logic."""
    from dvd_gan_amd import lib as L
    lib = L.lib()

    def conv(C_, cout, k, H, W, T=1, kt=1, frames=64, **kw):
        d = L.ConvDesc()
        d.dtype, d.frames, d.T, d.H, d.W = L.BF16, frames, T, H, W
        d.C, d.ldi, d.Cout, d.ldo = C_, C_, cout, max(8, (cout + 7) // 8 * 8)
        d.kt, d.kh, d.kw, d.nsplit = kt, k, k, 1
        d.inp = d.out = d.w = 1                      # never dereferenced by the query
        for key, v in kw.items():
            setattr(d, key, v)
        return d

    want = lambda d: lib.dvd_conv_wants_fragment_major(ctypes.byref(d))
    assert want(conv(256, 256, 5, 32, 32)) == 1               # halo-staged kernel, fragment-major image
    assert want(conv(8, 64, 3, 64, 64)) == 2                  # a discriminator stem: thin-input image
    assert want(conv(8, 64, 3, 32, 32, T=12, kt=3)) == 2
    assert want(conv(64, 3, 3, 64, 64)) == 3                  # the RGB layer: thin-output image
    assert want(conv(8, 64, 3, 64, 64, out_f32=1)) == 1       # fp32 output: the general kernel
    assert want(conv(8, 64, 3, 128, 128)) == 1                # 128-pixel lines: the general kernel
    assert want(conv(128, 128, 1, 64, 64)) == 0               # 1 x 1: tap-by-tap kernel, no image
    d = conv(8, 64, 3, 64, 64)
    d.wq, d.wq_kind = 1, 3                                    # a thin-output image handed to a thin-input request
    assert lib.dvd_conv_forward(ctypes.byref(d), None) == -1
    ok = lib.dvd_attention_mfma_ok
    assert ok(L.BF16, 160, 16, 16, 32, 128, 128, 256) == 1 and ok(L.BF16, 160, 16, 16, 32, 128, 128, 1024) == 1
    assert ok(L.F32, 160, 16, 16, 32, 128, 128, 256) == 0 and ok(L.BF16, 80, 8, 8, 16, 64, 64, 256) == 0
    assert ok(L.BF16, 160, 16, 16, 32, 128, 128, 48) == 0     # not whole 32-token blocks
    w = L.WgradDesc()
    w.dtype, w.frames, w.T, w.H, w.W = L.BF16, 64, 12, 32, 32
    w.C, w.ldx, w.Cin_real, w.Cout, w.Cy, w.ldy = 8, 8, 3, 64, 64, 64
    w.kt, w.kh, w.kw = 3, 3, 3
    w.s_co, w.s_ci, w.s_tap = 81, 27, 1
    w.x = w.dy = w.dw = 1
    thin = lib.dvd_conv_wgrad_ws_floats(ctypes.byref(w))
    assert thin == (2048 + 16) * (18 * 1024 + 8)              # persistent workgroups + the 16 partials of the first reduce stage
    w.dtype = L.F32
    assert lib.dvd_conv_wgrad_ws_floats(ctypes.byref(w)) != thin     # exact mode: the general kernels
    ws = lib.dvd_convgru_ws_floats
    assert ws(L.BF16, 64, 4, 4, 256, 3) == 8 * 1024 * 512     # 8 slices of [1024 rows][2h = 512]
    assert ws(L.BF16, 64, 32, 32, 128, 3) == 65536 * 256      # unsplit: one slab of whole 256-column tiles


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
