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
    # ... and nothing else: the internal cross-file entry points (common.h) have hidden visibility
    import shutil
    import subprocess
    if shutil.which("nm"):
        out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH]).decode()
        # EVERY defined text symbol, whatever its name (C++ helpers and kernel stubs included): -fvisibility=hidden + the
        # header's visibility pragma leave exactly the declared entry points (_init / _fini come from the C runtime)
        exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1] not in ("_init", "_fini"))
        assert exported == names, sorted(set(exported) ^ set(names))


def test_descriptor_layouts_match_the_library():
    """Layout handshake: sizeof() of every descriptor struct as the .so was compiled == the ctypes mirror in
    dvd_gan_amd/lib.py == (for dvd_conv_desc) the stub INTEGRATION.md section 2 prints.  Changing a descriptor in the header without
    touching lib.py or the document fails here, on the CPU."""
    import ctypes as C
    from dvd_gan_amd import lib as L
    lib = L.lib()
    assert set(L.STRUCT_MIRRORS) == {0, 1, 2, 3, 4, 5, 6}
    for which, mirror in L.STRUCT_MIRRORS.items():
        assert lib.dvd_struct_size(which) == C.sizeof(mirror), (which, mirror.__name__)
    assert lib.dvd_struct_size(99) == -1
    # the struct literally as INTEGRATION.md prints it
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = doc.index("class ConvDesc(C.Structure):")
    b = doc.index("assert lib.dvd_abi_version()", a)
    ns = {"C": C}
    exec(doc[a:b], ns)
    stub = ns["ConvDesc"]
    assert C.sizeof(stub) == lib.dvd_struct_size(0)
    assert [(n, getattr(stub, n).offset) for n, _ in stub._fields_] == [(n, getattr(L.ConvDesc, n).offset) for n, _ in L.ConvDesc._fields_]
    assert "dvd_abi_version() == %d" % L.ABI_VERSION in doc
    # a mirror that lost its tail is refused at load
    saved = L.STRUCT_MIRRORS[0]

    class Short(C.Structure):
        _fields_ = L.ConvDesc._fields_[:-2]
    try:
        L.STRUCT_MIRRORS[0] = Short
        L._lib = None
        with pytest.raises(RuntimeError, match="sizeof descriptor 0"):
            L.lib()
    finally:
        L.STRUCT_MIRRORS[0] = saved
        L._lib = None
        L.lib()


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
    assert want(conv(8, 64, 3, 64, 64, out_f32=1)) == 2       # fp32 output (parity tests): the thin-input kernel as well
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


def test_convgru_stack_queries_need_no_gpu():
    """dvd_convgru_stack_ok / dvd_convgru_stack_ws_floats (layer wavefront over a ConvGRU, Module/ConvGRU.py:57-133) are pure host
    logic: which stacks the grouped-launch path serves, and how much split-K slab space its schedule needs."""
    import ctypes as C
    from dvd_gan_amd import lib as L
    lib = L.lib()

    def stack(S, hids, ks, dtype=L.BF16, B=64, T=48, cin0=256):
        sd = L.GruStackDesc()
        sd.n_layers = len(hids)
        for l, (h, k) in enumerate(zip(hids, ks)):
            d = sd.layer[l]
            d.dtype, d.T, d.B, d.H, d.W, d.hidden, d.k = dtype, T, B, S, S, h, k
            d.gx_stride = B * S * S * 3 * h
            for f in ("gx", "w_ur", "w_o", "w_ur_q", "w_o_q", "wd_ur", "wd_o", "wd_ur_q", "wd_o_q", "h_all", "u_all", "r_all", "o_all",
                      "hr_all", "tickets", "dg", "carry"):
                setattr(d, f, 1)                     # never dereferenced by the queries
            if l:
                sd.cin[l] = hids[l - 1]
                sd.wx[l] = sd.wx_q[l] = sd.bx[l] = sd.wdx[l] = sd.wdx_q[l] = sd.dh_mid[l] = 1
        sd.ws = 1
        return sd

    ok = lambda sd, bwd=0: lib.dvd_convgru_stack_ok(C.byref(sd), bwd)
    g3 = stack(32, [128, 256, 128], [3, 5, 5])
    assert ok(g3) == 1 and ok(g3, 1) == 1
    assert ok(stack(4, [256, 512, 256], [3, 5, 3])) == 1 and ok(stack(8, [256, 512, 256], [3, 5, 3])) == 1
    assert ok(stack(12, [256, 512, 256], [3, 5, 3])) == 0          # not a power of two
    assert ok(stack(32, [128, 256, 128], [3, 7, 5])) == 0          # 7 x 7 taps
    assert ok(stack(32, [128, 256, 128], [3, 5, 5], dtype=L.F32)) == 0
    bad = stack(32, [128, 256, 128], [3, 5, 5])
    bad.wx_q[1] = None
    assert ok(bad) == 0 and ok(bad, 1) == 1                        # the forward pass needs the x-part image, the backward pass does not
    assert lib.dvd_convgru_stack_ws_floats(C.byref(g3)) >= 1       # one-round groups at 32 x 32: nothing is split
    small = lib.dvd_convgru_stack_ws_floats(C.byref(stack(4, [256, 512, 256], [3, 5, 3])))
    assert small > 16384 and small % 16384 == 0                    # 4 x 4 frames: split-K slabs of whole 128 x 128 tiles
    # the sizing query walks exactly the schedule of the launch: repeatable, and a supplied initial state (more members in the
    # pipeline-fill groups) changes the plan without ever yielding less than one float
    for S in (4, 8, 16):
        sd = stack(S, [256, 512, 256], [3, 5, 3])
        a = lib.dvd_convgru_stack_ws_floats(C.byref(sd))
        assert a == lib.dvd_convgru_stack_ws_floats(C.byref(sd)) and a >= 1
        for l in range(3):
            sd.layer[l].h0 = 1
        b = lib.dvd_convgru_stack_ws_floats(C.byref(sd))
        assert b >= 1 and b % 16384 == 0
    assert lib.dvd_convgru_stack_forward(C.byref(stack(12, [256], [3])), None) == -2


def test_hot_kernels_use_no_scratch_memory():
    """hipcc's own resource report of every kernel (dvd_gan_amd/csrc/build.sh keeps it as build/<file>.res): the convolution, weight-
    gradient, ConvGRU and attention kernels must keep their accumulators in registers -- no scratch, no spilled vector registers.
    (Round 6: a runtime-indexed accumulator array in a new epilogue branch put EVERY convolution kernel's accumulators into scratch;
    parity stayed green, the step ran 3x slower.)  Known exception: a few scalar-register spills in rarely used 5-tap ReLU variants."""
    import glob
    build = os.path.join(ROOT, "dvd_gan_amd", "csrc", "build")
    files = sorted(glob.glob(os.path.join(build, "*.res")))
    if not files:
        pytest.skip("no build/*.res: the library was not built by csrc/build.sh in this tree")
    hot = ("conv_halo", "conv_group", "conv_igemm", "conv_wgrad", "conv_thin", "wgrad_thin", "attn_", "gru_", "cbn_", "bn_stats")
    seen, bad = 0, []
    for path in files:
        name, rec = None, {}
        for line in open(path):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                name, rec = m.group(1), {}
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
            if m and name:
                rec[m.group(1).strip()] = int(m.group(2))
                if m.group(1).strip() == "VGPRs Spill" and any(h in name for h in hot):
                    seen += 1
                    if rec.get("ScratchSize", 0) != 0 and rec.get("SGPRs Spill", 0) == 0 or rec["VGPRs Spill"] != 0:
                        bad.append((os.path.basename(path), name, rec))
    assert seen >= 100, seen
    assert not bad, bad[:5]
