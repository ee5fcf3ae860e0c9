#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference; it never travels to the
GPU box).  Every fixture stores explicit inputs, the full initial state_dict
(including SN u/v and BN buffers), outputs, selected gradients and the
post-forward state, so nothing depends on cross-version RNG streams.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixture ids follow SURVEY.md section 8(c): F1..F10.
Key scheme inside each .npz:   in.<name>  sd0.<state_dict key>  out.<name>
                               grad.<param key>  sd1.<state_dict key>  meta.<name>
"""
import os
import sys
import types
import argparse
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    """Stub the two imports the image lacks, then import the reference modules."""
    if "tensorboardX" not in sys.modules:
        tb = types.ModuleType("tensorboardX")
        tb.SummaryWriter = type("SummaryWriter", (), {})
        sys.modules["tensorboardX"] = tb
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.save_image = lambda *a, **k: None
        tvu.make_grid = lambda x, *a, **k: x
        tv.utils = tvu
        tv.get_image_backend = lambda: "PIL"
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu
    if REF not in sys.path:
        sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    import Module.Normalization as N
    import Module.ConvGRU as CG
    import Module.GResBlock as GR
    import Module.Generator as GE
    import Module.Discriminators as DI
    import Module.Attention as AT
    import utils as U
    return types.SimpleNamespace(N=N, CG=CG, GR=GR, GE=GE, DI=DI, AT=AT, U=U)


def npy(t):
    return t.detach().cpu().numpy().copy()


def put_sd(store, tag, module):
    for k, v in module.state_dict().items():
        store[f"{tag}.{k}"] = npy(v)


def put_state(store, tag, module):
    """Only what a forward pass mutates: SN u/v and BN buffers."""
    for k, v in module.state_dict().items():
        if k.endswith(("weight_u", "weight_v", "running_mean", "running_var", "num_batches_tracked")):
            store[f"{tag}.{k}"] = npy(v)


def put_grads(store, module, only=None):
    for k, p in module.named_parameters():
        if p.grad is not None and (only is None or k in only):
            store[f"grad.{k}"] = npy(p.grad)


def save(name, store):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(store)} arrays)")


# ----------------------------------------------------------------------------- F1
def f1_spectral_norm(R):
    """SpectralNorm on Conv2d / Conv3d / Linear(->1) / Embedding: u, v, sigma-normalised
    weight after 1 and 3 forwards, gradient wrt weight_bar (Normalization.py:10-64)."""
    torch.manual_seed(101)
    import torch.nn as nn
    st = {}
    cases = {
        "conv2d": (nn.Conv2d(4, 8, 3, padding=1), torch.randn(2, 4, 5, 5)),
        "conv3d": (nn.Conv3d(3, 6, 3, padding=1), torch.randn(2, 3, 4, 5, 5)),
        "linear": (nn.Linear(16, 1), torch.randn(5, 16)),
        "embed": (nn.Embedding(7, 16), torch.tensor([0, 3, 6, 3])),
    }
    for name, (inner, x) in cases.items():
        sn = R.N.SpectralNorm(inner)
        put_sd(st, f"{name}.sd0", sn)
        st[f"{name}.in.x"] = npy(x)
        y = sn(x)
        st[f"{name}.out.y1"] = npy(y)
        st[f"{name}.out.w1"] = npy(sn.module.weight)
        gy = torch.randn_like(y)
        st[f"{name}.in.gy"] = npy(gy)
        y.backward(gy)
        for k, p in sn.named_parameters():
            if p.grad is not None:
                st[f"{name}.grad.{k}"] = npy(p.grad)
        put_sd(st, f"{name}.sd1", sn)
        sn(x)
        y3 = sn(x)
        st[f"{name}.out.y3"] = npy(y3)
        put_sd(st, f"{name}.sd3", sn)
    save("f1_spectral_norm", st)


# ----------------------------------------------------------------------------- F2
def f2_conditional_norm(R):
    """ConditionalNorm train + eval (Normalization.py:66-88)."""
    torch.manual_seed(102)
    st = {}
    cn = R.N.ConditionalNorm(6, 10)
    x = torch.randn(5, 6, 4, 4, requires_grad=True)
    c = torch.randn(5, 10, requires_grad=True)
    put_sd(st, "sd0", cn)
    st["in.x"], st["in.cond"] = npy(x), npy(c)
    cn.train()
    y = cn(x, c)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["in.gy"] = npy(gy)
    st["out.y_train"] = npy(y)
    st["grad.x"], st["grad.cond"] = npy(x.grad), npy(c.grad)
    put_grads(st, cn)
    put_sd(st, "sd1", cn)
    cn.eval()
    st["out.y_eval"] = npy(cn(x, c))
    save("f2_conditional_norm", st)


# ----------------------------------------------------------------------------- F3
def f3_gresblock(R):
    """GResBlock with upsample_factor 1 and 2, fwd + bwd (GResBlock.py:42-86)."""
    torch.manual_seed(103)
    st = {}
    for tag, up, cin, cout in (("up1", 1, 8, 8), ("up2", 2, 8, 4)):
        blk = R.GR.GResBlock(cin, cout, n_class=12, upsample_factor=up)
        x = torch.randn(6, cin, 4, 4, requires_grad=True)
        cond = torch.randn(6, 12, requires_grad=True)
        put_sd(st, f"{tag}.sd0", blk)
        st[f"{tag}.in.x"], st[f"{tag}.in.cond"] = npy(x), npy(cond)
        y = blk(x, cond)
        gy = torch.randn_like(y)
        y.backward(gy)
        st[f"{tag}.in.gy"], st[f"{tag}.out.y"] = npy(gy), npy(y)
        st[f"{tag}.grad.x"], st[f"{tag}.grad.cond"] = npy(x.grad), npy(cond.grad)
        for k, p in blk.named_parameters():
            if p.grad is not None:
                st[f"{tag}.grad.{k}"] = npy(p.grad)
        put_sd(st, f"{tag}.sd1", blk)
    save("f3_gresblock", st)


# ----------------------------------------------------------------------------- F4
def f4_convgru(R):
    """ConvGRUCell and 3-layer ConvGRU over T=4, kernels (3,5,5) fwd + BPTT (ConvGRU.py:29-133)."""
    torch.manual_seed(104)
    st = {}
    cell = R.CG.ConvGRUCell(8, 16, 5)
    for p in cell.parameters():  # non-zero biases so the bias path is exercised
        if p.dim() == 1:
            p.data.normal_(0, 0.1)
    x = torch.randn(3, 8, 4, 8, requires_grad=True)       # H != W on purpose
    h = torch.randn(3, 16, 4, 8, requires_grad=True)
    put_sd(st, "cell.sd0", cell)
    st["cell.in.x"], st["cell.in.h"] = npy(x), npy(h)
    y0 = cell(x)           # prev_state None -> zeros (ConvGRU.py:31-44)
    y = cell(x, h)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["cell.out.y_h0"], st["cell.out.y"], st["cell.in.gy"] = npy(y0), npy(y), npy(gy)
    st["cell.grad.x"], st["cell.grad.h"] = npy(x.grad), npy(h.grad)
    for k, p in cell.named_parameters():
        st[f"cell.grad.{k}"] = npy(p.grad)

    gru = R.CG.ConvGRU(8, hidden_sizes=[8, 16, 8], kernel_sizes=[3, 5, 5], n_layers=3)
    for p in gru.parameters():
        if p.dim() == 1:
            p.data.normal_(0, 0.1)
    T = 4
    xs = torch.randn(T, 2, 8, 8, 4, requires_grad=True)
    put_sd(st, "gru.sd0", gru)
    st["gru.in.xs"] = npy(xs)
    hidden, outs = None, []
    for t in range(T):
        hidden = gru(xs[t], hidden)
        outs.append(hidden[-1])
    y = torch.stack(outs)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["gru.out.y"], st["gru.in.gy"], st["gru.grad.xs"] = npy(y), npy(gy), npy(xs.grad)
    for li, hl in enumerate(hidden):
        st[f"gru.out.h_last.{li}"] = npy(hl)
    for k, p in gru.named_parameters():
        st[f"gru.grad.{k}"] = npy(p.grad)
    save("f4_convgru", st)


# ----------------------------------------------------------------------------- F5
def f5_attention(R):
    """2-D SelfAttention of the discriminators, gamma != 0 (Discriminators.py:82-119);
    plus the (defined, never invoked) 5-D SelfAttention / SeparableAttn (Attention.py)."""
    torch.manual_seed(105)
    st = {}
    for tag, C, S in (("n16", 16, 4), ("n64", 8, 8)):
        at = R.DI.SelfAttention(C)
        at.gamma.data.fill_(0.7)
        for p in (at.query_conv.bias, at.key_conv.bias, at.value_conv.bias):
            p.data.normal_(0, 0.1)
        x = torch.randn(3, C, S, S, requires_grad=True)
        put_sd(st, f"{tag}.sd0", at)
        st[f"{tag}.in.x"] = npy(x)
        y = at(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        st[f"{tag}.out.y"], st[f"{tag}.in.gy"], st[f"{tag}.grad.x"] = npy(y), npy(gy), npy(x.grad)
        for k, p in at.named_parameters():
            st[f"{tag}.grad.{k}"] = npy(p.grad)
    a3 = R.AT.SelfAttention(8)
    a3.gamma.data.fill_(0.5)
    x = torch.randn(2, 8, 4, 4, 4, requires_grad=True)
    put_sd(st, "attn3d.sd0", a3)
    st["attn3d.in.x"] = npy(x)
    y = a3(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["attn3d.out.y"], st["attn3d.in.gy"], st["attn3d.grad.x"] = npy(y), npy(gy), npy(x.grad)
    for k, p in a3.named_parameters():
        st[f"attn3d.grad.{k}"] = npy(p.grad)
    sp = R.AT.SeparableAttn(8)
    for m in sp.modules():
        if hasattr(m, "gamma"):
            m.gamma.data.fill_(0.3)
    x = torch.randn(2, 8, 4, 4, 4, requires_grad=True)
    put_sd(st, "sep.sd0", sp)
    st["sep.in.x"] = npy(x)
    y = sp(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["sep.out.y"], st["sep.in.gy"], st["sep.grad.x"] = npy(y), npy(gy), npy(x.grad)
    for k, p in sp.named_parameters():
        st[f"sep.grad.{k}"] = npy(p.grad)
    save("f5_attention", st)


# ----------------------------------------------------------------------------- F6
def f6_generator(R):
    """Generator ch=2: (B=3,T=4: B does not divide T) and (B=2,T=4: divides) pin the
    condition mis-ordering quirk (Generator.py:103-110); latent_dim 2 and 4.
    Case a stores every gradient; case b stores the output and a few gradients."""
    st = {}
    few = ("conv.0.cells.0.update_gate.weight", "conv.4.conv0.module.weight_bar",
           "conv.11.CBNorm2.embed.weight", "embedding.weight", "affine_transfrom.bias")
    for tag, B, T, ld, seed, full in (("a", 3, 4, 2, 106, True), ("b", 2, 4, 4, 107, False)):
        torch.manual_seed(seed)
        G = R.GE.Generator(in_dim=12, latent_dim=ld, n_class=3, ch=2, n_frames=T)
        for k, p in G.named_parameters():  # exercise the bias paths of the GRU gates
            if "gate.bias" in k:
                p.data.normal_(0, 0.05)
        z = torch.randn(B, 12)
        cls = torch.randint(0, 3, (B,))
        put_sd(st, f"{tag}.sd0", G)
        st[f"{tag}.in.z"], st[f"{tag}.in.cls"] = npy(z), npy(cls)
        G.train()
        y = G(z, cls)
        gy = torch.randn_like(y)
        y.backward(gy)
        st[f"{tag}.out.y"], st[f"{tag}.in.gy"] = npy(y), npy(gy)
        for k, p in G.named_parameters():
            if p.grad is not None and (full or k in few):
                st[f"{tag}.grad.{k}"] = npy(p.grad)
        put_state(st, f"{tag}.sd1", G)
        G.eval()
        with torch.no_grad():
            st[f"{tag}.out.y_eval"] = npy(G(z, cls))
        put_state(st, f"{tag}.sd2", G)
    save("f6_generator", st)


# ----------------------------------------------------------------------------- F7
def f7_discriminators(R):
    """SpatialDiscriminator / TemporalDiscriminator chn=2 fwd + bwd at 64x64
    (+ D_s at 32x32; D_t at 32x32 frames is an expected RuntimeError)."""
    st = {}
    torch.manual_seed(108)
    Ds = R.DI.SpatialDiscriminator(chn=2, n_class=3)
    Ds.attn.gamma.data.fill_(0.4)
    x = (torch.rand(2, 3, 3, 64, 64) * 2 - 1).requires_grad_(True)   # [B,k,3,H,W]
    cls = torch.tensor([2, 0])
    put_sd(st, "ds.sd0", Ds)
    st["ds.in.x"], st["ds.in.cls"] = npy(x), npy(cls)
    y = Ds(x, cls)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["ds.out.y"], st["ds.in.gy"], st["ds.grad.x"] = npy(y), npy(gy), npy(x.grad)
    for k, p in Ds.named_parameters():
        if p.grad is not None:
            st[f"ds.grad.{k}"] = npy(p.grad)
    put_sd(st, "ds.sd1", Ds)
    x32 = torch.rand(2, 2, 3, 32, 32) * 2 - 1
    st["ds32.in.x"] = npy(x32)
    with torch.no_grad():
        st["ds32.out.y"] = npy(Ds(x32, cls))

    torch.manual_seed(109)
    Dt = R.DI.TemporalDiscriminator(chn=2, n_class=3)
    Dt.self_attn.gamma.data.fill_(-0.6)
    x = (torch.rand(2, 3, 8, 32, 32) * 2 - 1).requires_grad_(True)   # [B,3,T,H/2,W/2]
    put_sd(st, "dt.sd0", Dt)
    st["dt.in.x"], st["dt.in.cls"] = npy(x), npy(cls)
    y = Dt(x, cls)
    gy = torch.randn_like(y)
    y.backward(gy)
    st["dt.out.y"], st["dt.in.gy"], st["dt.grad.x"] = npy(y), npy(gy), npy(x.grad)
    for k, p in Dt.named_parameters():
        if p.grad is not None:
            st[f"dt.grad.{k}"] = npy(p.grad)
    put_sd(st, "dt.sd1", Dt)
    try:
        Dt(torch.rand(1, 3, 8, 16, 16), cls[:1])
        st["dt16.meta.raises"] = np.array(0)
    except RuntimeError:
        st["dt16.meta.raises"] = np.array(1)
    save("f7_discriminators", st)


# ----------------------------------------------------------------------------- F8
def f8_helpers(R):
    """sample_k_frames with recorded frame ids, vid_downsample (utils.py:60-63,77-83)."""
    torch.manual_seed(110)
    st = {}
    data = torch.randn(2, 6, 3, 8, 8)
    st["in.data"] = npy(data)
    rec = {}
    orig = torch.randperm

    def spy(n, *a, **k):
        r = orig(n, *a, **k)
        rec["perm"] = r.clone()
        return r
    torch.randperm = spy
    try:
        out = R.U.sample_k_frames(data, 6, 4)
        st["in.perm"], st["out.sample_k4"] = npy(rec["perm"]), npy(out)
        out = R.U.sample_k_frames(data, 6, 9)      # k > T -> every frame, sorted
        st["in.perm_k9"], st["out.sample_k9"] = npy(rec["perm"]), npy(out)
    finally:
        torch.randperm = orig
    st["out.down"] = npy(R.U.vid_downsample(data))
    save("f8_helpers", st)


# ----------------------------------------------------------------------------- F9 / F10
def conditional_generator_forward(G, hid):
    """-> a forward(z, class_id) for the REFERENCE generator `G` that does what Generator.forward (Generator.py:63-120) does with
    ONE change: the first frame of ConvGRU number n is evaluated as `conv(x_0, hid[n])` instead of `conv(x_0)` -- the hook
    ConvGRU.forward(x, hidden) offers (ConvGRU.py:104-118); frame-conditional variant, BASELINE configs[4].  Every layer is
    the reference's own module; with hid[n] = None the result is that of the unmodified forward (asserted by the caller)."""
    import torch.nn.functional as F

    def forward(z, class_id):
        T = G.n_frames
        cond = torch.cat((z, G.embedding(class_id)), dim=1)
        y = G.affine_transfrom(cond).view(-1, 8 * G.ch, G.latent_dim, G.latent_dim)
        n = 0
        for k, m in enumerate(G.conv):
            if hasattr(m, "cells"):                                  # ConvGRU
                if k > 0:
                    y = y.view(-1, T, *y.shape[1:])
                state, outs = hid[n], []
                n += 1
                for i in range(T):
                    state = m(y if k == 0 else y[:, i], state)
                    outs.append(state[-1])
                y = torch.stack(outs, 1)
                y = y.reshape(-1, *y.shape[2:])
            else:                                                    # GResBlock: condition rows t-major (quirk 1)
                y = m(y, cond.repeat(T, 1))
        y = torch.tanh(G.colorize(F.relu(y)))
        return y.view(-1, T, *y.shape[1:])
    return forward


def run_trainer(R, *, adv_loss, ch, T, k, B, n_class, steps, seed, z_dim=120, lr=5e-5,
                grads_of=(), n_batches=None, synth_big=False, grad_head=None, d_iters=1, bf16_state=False,
                latent_dim=4, hidden=False):
    """Drive the unmodified reference Trainer.train() (trainer.py:189-307) on CPU and
    record every RNG draw, the six loss terms per step, named gradients and parameter
    checksums at each optimizer.step().
    synth_big: every floating-point state tensor with >= synth.BIG elements and the input clips are set to the
    closed-form values of tests/golden/synth.py BEFORE the reference runs, and are not stored.
    grad_head: store only the first `grad_head` elements (flattened) of each named gradient.
    bf16_state: the reference's OWN default initialisation (orthogonal ConvGRU weights, ConvGRU.py:20-26, xavier attention
    projections, N(0,1) SN vectors ...) and the input clips are rounded to bf16-representable values BEFORE the reference
    runs, and stored as 2-byte values ('<tag>.sd0b.<key>', 'in.realb.<i>': the upper halves of the float32 words) -- a bf16
    implementation then starts from exactly the reference's operands, and the fixture stays small.
    d_iters > 1 (trainer.py:230): the draws of discriminator iteration i of step s are stored as in.<name>.<s>.<i>; losses,
    gradients and checksums are those of the LAST discriminator iteration of the step (and of the generator update).
    latent_dim != 4 (frames of 16 * latent_dim pixels): trainer.py:349 never passes latent_dim, so the Trainer's generator is
    replaced by the reference Generator(latent_dim=...) and `select_opt_schr` re-run before training (meta.latent_dim).
    hidden: initial ConvGRU states (closed forms of synth.py, `hidden.<gru>.<layer>`, scale 0.5) supplied at the first frame
    of every ConvGRU through `conditional_generator_forward`; their gradients at the generator update are stored as
    `hgrad.<step>.<gru>.<layer>` (head) and `out.hgsum.<step>` (|.| checksums)."""
    import trainer as TR
    import synth
    import torch.nn as nn
    nn.Module.cuda = lambda self, *a, **kw: self          # trainer.py:349-351 call .cuda()
    torch.manual_seed(seed)
    cfg = argparse.Namespace(
        model="dvd-gan", adv_loss=adv_loss, imsize=64, g_num=5, z_dim=z_dim, g_chn=ch, ds_chn=ch,
        dt_chn=ch, n_frames=T, g_conv_dim=64, d_conv_dim=64, lr_schr="const", lambda_gp=10,
        total_epoch=1 if n_batches is None else steps // n_batches, d_iters=d_iters, g_iters=1, batch_size=B, num_workers=0, g_lr=lr, d_lr=lr,
        lr_decay=0.9999, beta1=0.0, beta2=0.9, pretrained_model=None, n_class=n_class,
        k_sample=k, dataset="ucf101", use_tensorboard=False, test_batch_size=1,
        image_path="/tmp/x", log_path="/tmp/x", model_save_path="/tmp/x", sample_path="/tmp/x",
        log_epoch=10 ** 6, sample_epoch=10 ** 6, model_save_epoch=10 ** 6, version="g",
        gpus=[], parallel=False)
    gen = torch.Generator().manual_seed(seed + 1)
    nb_ = steps if n_batches is None else n_batches
    FR = 16 * latent_dim
    if synth_big:
        loader = [(torch.from_numpy(synth.uniform(f"real.{i}", (B, 3, T, FR, FR))),
                   torch.randint(0, n_class, (B,), generator=gen)) for i in range(nb_)]
    else:
        loader = [((torch.rand(B, 3, T, FR, FR, generator=gen) * 2 - 1),
                   torch.randint(0, n_class, (B,), generator=gen)) for _ in range(nb_)]
    if bf16_state:
        loader = [(v.to(torch.bfloat16).float(), l) for v, l in loader]
    tr = TR.Trainer(loader, cfg)
    if latent_dim != 4:
        tr.G = R.GE.Generator(z_dim, latent_dim, n_class=n_class, ch=ch, n_frames=T)
        tr.select_opt_schr()
    hid = None
    if hidden:
        hid = []
        for gi, m in enumerate(mm for mm in tr.G.conv if hasattr(mm, "cells")):
            S_ = latent_dim << gi
            hid.append([torch.from_numpy(synth.uniform(f"hidden.{gi}.{l}", (B, m.hidden_sizes[l], S_, S_), 0.5)).requires_grad_(True)
                        for l in range(m.n_layers)])
        with torch.no_grad():                    # the restated forward equals the reference's own when no state is supplied
            import copy
            Gc = copy.deepcopy(tr.G)
            zz, cc = torch.randn(B, z_dim, generator=gen), torch.randint(0, n_class, (B,), generator=gen)
            a = copy.deepcopy(Gc)(zz, cc)
            b = conditional_generator_forward(Gc, [None] * len(hid))(zz, cc)
            assert torch.equal(a, b), "conditional_generator_forward deviates from Generator.forward"
        tr.G.forward = conditional_generator_forward(tr.G, hid)
    st = {}
    for net, tag in ((tr.G, "G"), (tr.D_s, "Ds"), (tr.D_t, "Dt")):
        for kk, v in net.state_dict().items():
            if bf16_state and v.is_floating_point():
                v.copy_(v.to(torch.bfloat16).float())
                st[f"{tag}.sd0b.{kk}"] = v.to(torch.bfloat16).view(torch.int16).numpy().copy()
            elif synth_big and v.is_floating_point() and v.numel() >= synth.BIG:
                v.copy_(torch.from_numpy(synth.weight(tag + "." + kk, tuple(v.shape))))    # state_dict tensors alias the parameters
            else:
                st[f"{tag}.sd0.{kk}"] = npy(v)
    for i, (v, l) in enumerate(loader):
        if bf16_state:
            st[f"in.realb.{i}"] = v.to(torch.bfloat16).view(torch.int16).numpy().copy()
        elif not synth_big:
            st[f"in.real.{i}"] = npy(v)
        st[f"in.labels.{i}"] = npy(l)

    draws = {"randperm": [], "randn": [], "randint": []}
    o_perm, o_randn, o_randint = torch.randperm, torch.randn, torch.randint

    def w_perm(*a, **kw):
        r = o_perm(*a, **kw); draws["randperm"].append(npy(r)); return r

    def w_randn(*a, **kw):
        r = o_randn(*a, **kw); draws["randn"].append(npy(r)); return r

    def w_randint(*a, **kw):
        r = o_randint(*a, **kw); draws["randint"].append(npy(r)); return r
    losses = []
    o_calc = tr.calc_loss

    def w_calc(x, flag):
        r = o_calc(x, flag); losses.append(float(r)); return r
    tr.calc_loss = w_calc
    snaps = {"Ds": [], "Dt": [], "G": []}
    hgrads = []

    def wrap_opt(opt, net, tag):
        o_step = opt.step

        def stepper(*a, **kw):
            g = {kk: (npy(p.grad) if grad_head is None else npy(p.grad.reshape(-1)[:grad_head]))
                 for kk, p in net.named_parameters() if p.grad is not None and kk in grads_of}
            gsum = {kk: float(p.grad.double().abs().sum()) for kk, p in net.named_parameters()
                    if p.grad is not None}
            if tag == "G" and hid is not None:
                hgrads.append([[npy(h.grad) for h in hs] for hs in hid])
                for hs in hid:
                    for h in hs:
                        h.grad = None
            r = o_step(*a, **kw)
            psum = {kk: float(v.double().abs().sum()) for kk, v in net.state_dict().items()}
            snaps[tag].append((g, gsum, psum))
            return r
        opt.step = stepper
    wrap_opt(tr.ds_optimizer, tr.D_s, "Ds")
    wrap_opt(tr.dt_optimizer, tr.D_t, "Dt")
    wrap_opt(tr.g_optimizer, tr.G, "G")
    torch.randperm, torch.randn, torch.randint = w_perm, w_randn, w_randint
    try:
        tr.train()
    finally:
        torch.randperm, torch.randn, torch.randint = o_perm, o_randn, o_randint
    # draws[randn][0] is fixed_z (trainer.py:195); per step afterwards: perm, randn, randint, perm
    st["in.fixed_z"] = draws["randn"][0]
    D = d_iters
    for s in range(steps):
        if D == 1:
            st[f"in.perm_real.{s}"] = draws["randperm"][2 * s]
            st[f"in.perm_fake.{s}"] = draws["randperm"][2 * s + 1]
            st[f"in.z.{s}"] = draws["randn"][1 + s]
            st[f"in.z_class.{s}"] = draws["randint"][s]
        else:
            for i in range(D):
                st[f"in.perm_real.{s}.{i}"] = draws["randperm"][2 * (s * D + i)]
                st[f"in.perm_fake.{s}.{i}"] = draws["randperm"][2 * (s * D + i) + 1]
                st[f"in.z.{s}.{i}"] = draws["randn"][1 + s * D + i]
                st[f"in.z_class.{s}.{i}"] = draws["randint"][s * D + i]
        per = 4 * D + 2                       # calc_loss calls per step: 4 per discriminator iteration + 2
        last = losses[per * s + 4 * (D - 1): per * s + 4 * D] + losses[per * s + 4 * D: per * (s + 1)]
        st[f"out.losses.{s}"] = np.array(last, dtype=np.float64)
        for tag in ("Ds", "Dt", "G"):
            g, gsum, psum = snaps[tag][s if tag == "G" else s * D + D - 1]
            for kk, v in g.items():
                st[f"grad.{s}.{tag}.{kk}"] = v
            keys = sorted(gsum)
            st[f"meta.gsum_keys.{tag}"] = np.array(keys)
            st[f"out.gsum.{s}.{tag}"] = np.array([gsum[kk] for kk in keys])
            keys = sorted(psum)
            st[f"meta.psum_keys.{tag}"] = np.array(keys)
            st[f"out.psum.{s}.{tag}"] = np.array([psum[kk] for kk in keys])
    for s, hg in enumerate(hgrads):
        sums = []
        for gi, hs in enumerate(hg):
            for l, v in enumerate(hs):
                st[f"hgrad.{s}.{gi}.{l}"] = v if grad_head is None else v.reshape(-1)[:grad_head]
                sums.append(float(np.abs(v.astype(np.float64)).sum()))
        st[f"out.hgsum.{s}"] = np.array(sums)
    st["meta.latent_dim"] = np.array(latent_dim)
    st["meta.hidden"] = np.array(int(bool(hidden)))
    put_state(st, "G.sd1", tr.G)
    put_state(st, "Ds.sd1", tr.D_s)
    put_state(st, "Dt.sd1", tr.D_t)
    st["meta.cfg"] = np.array([ch, T, k, B, n_class, steps, z_dim], dtype=np.int64)
    st["meta.lr"] = np.array(lr)
    st["meta.synth"] = np.array(int(synth_big))
    st["meta.d_iters"] = np.array(d_iters)
    return st


def f9_trainer_steps(R):
    """Two full Trainer steps, hinge and wgan-gp, ch=2, T=8, k=4, B=2, 64x64.  Both runs use the
    same seed, hence the same initial state / inputs / RNG draws: the wgan-gp file keeps only
    its outputs and points at the hinge file for sd0 and the inputs."""
    names = ("conv.0.cells.1.update_gate.weight", "conv.9.cells.2.out_gate.weight",
             "conv.1.conv0.module.weight_bar", "conv.1.CBNorm1.embed.weight", "embedding.weight",
             "colorize.module.weight_bar", "pre_conv.0.module.weight_bar", "attn.gamma",
             "self_attn.gamma", "linear.module.weight_bar", "embed.module.weight_bar",
             "res3d.conv1.module.weight_bar", "conv1.conv_sc.module.weight_bar")
    ref = None
    for loss in ("hinge", "wgan-gp"):
        st = run_trainer(R, adv_loss=loss, ch=2, T=8, k=4, B=2, n_class=3, steps=2,
                         seed=120, z_dim=16, lr=2e-3, grads_of=names)
        if ref is None:
            ref = st
        else:
            for k in list(st):
                if ".sd0." in k or k.startswith("in."):
                    assert np.array_equal(st[k], ref[k]), k
                    del st[k]
        save("f9_trainer_" + loss.replace("-", ""), st)


def f10_config1(R):
    """Config-1 plumbing (frames must be 64x64; BASELINE configs[0]): T=16, B=2, n_class=1, k=8,
    ch=8, 3 steps, hinge, reference defaults lr=5e-5 / z_dim=120.  The large initial weights and the clips are the
    closed-form tensors of synth.py (8.5 M generator parameters would not fit a committed fixture).
    Losses + checksums only."""
    st = run_trainer(R, adv_loss="hinge", ch=8, T=16, k=8, B=2, n_class=1, steps=3, seed=140,
                     z_dim=120, lr=5e-5, n_batches=1, synth_big=True)   # one batch, re-iterated (trainer.py:216-221)
    keep = {k: v for k, v in st.items() if not (k.startswith("grad.") or ".sd1." in k)}
    save("f10_config1", keep)


F11_NAMES = ("conv.0.cells.1.update_gate.weight", "conv.3.cells.0.reset_gate.weight", "conv.6.cells.2.out_gate.bias",
             "conv.9.cells.2.out_gate.weight", "conv.9.cells.1.update_gate.bias",
             "conv.1.conv0.module.weight_bar", "conv.5.conv_sc.module.weight_bar", "conv.11.conv1.module.bias",
             "conv.1.CBNorm1.embed.weight", "conv.10.CBNorm2.embed.bias", "embedding.weight",
             "affine_transfrom.weight", "colorize.module.weight_bar",
             "pre_conv.0.module.weight_bar", "pre_conv.2.module.weight_bar", "pre_skip.module.bias", "attn.gamma",
             "self_attn.gamma", "attn.value_conv.weight", "self_attn.query_conv.weight", "linear.module.weight_bar",
             "embed.module.weight_bar", "res3d.conv1.module.weight_bar", "res3d.conv_sc.module.weight_bar",
             "conv1.conv_sc.module.weight_bar", "conv2.2.conv1.module.weight_bar", "conv.1.conv0.module.weight_bar")


def f11_full_width(R):
    """ONE step of the unmodified reference Trainer at the benchmark's per-clip shape (BASELINE configs[1]: ch=32, T=48,
    64x64, 101 classes, k=8, hinge, lr 5e-5) with B=2 -- the real 128..1024-channel widths.  Large weights and the clips
    are synth.py tensors; stored: small state, RNG draws, the six losses, |grad| checksums of EVERY parameter at its
    optimizer step and the first 8192 elements of the named gradients, post-step SN / BN state."""
    st = run_trainer(R, adv_loss="hinge", ch=32, T=48, k=8, B=2, n_class=101, steps=1, seed=150,
                     z_dim=120, lr=5e-5, grads_of=F11_NAMES, synth_big=True, grad_head=8192)
    keep = {k: v for k, v in st.items() if not (".sd1." in k and v.size >= 4096)}
    save("f11_full_width", keep)


def f13_two_d_iters(R):
    """d_iters = 2 (trainer.py:230): two discriminator updates per step with fresh draws, the generator forward twice in
    train mode, the generator update on the clips of the second iteration.  ch=2, T=8, k=4, B=1, 3 classes, 2 steps, hinge."""
    names = ("conv.0.cells.1.update_gate.weight", "conv.1.conv0.module.weight_bar", "embedding.weight",
             "colorize.module.weight_bar", "pre_conv.0.module.weight_bar", "linear.module.weight_bar",
             "res3d.conv1.module.weight_bar")
    st = run_trainer(R, adv_loss="hinge", ch=2, T=8, k=4, B=1, n_class=3, steps=2, seed=160, z_dim=12, lr=2e-3,
                     grads_of=names, d_iters=2)
    save("f13_two_d_iters", st)


F14_NAMES = ("conv.0.cells.1.update_gate.weight", "conv.3.cells.0.reset_gate.weight", "conv.9.cells.2.out_gate.weight",
             "conv.1.conv0.module.weight_bar", "conv.10.CBNorm2.embed.weight", "embedding.weight", "colorize.module.weight_bar",
             "pre_conv.0.module.weight_bar", "attn.value_conv.weight", "self_attn.query_conv.weight", "linear.module.weight_bar",
             "res3d.conv1.module.weight_bar", "conv1.conv_sc.module.weight_bar", "conv2.2.conv1.module.weight_bar")


def f14_default_init_bf16(R):
    """ONE step of the unmodified reference Trainer from ITS OWN default initialisation -- the state a fresh model starts
    from (orthogonal ConvGRU weights amplify perturbations over the recurrence far more than the uniform synth.py
    weights of F10 / F11 do) -- at ch=4, T=16, 64x64, B=2, k=4, 3 classes, hinge, lr 5e-5, z_dim 120.  Every weight,
    buffer and input clip is rounded to bf16 before the reference runs and stored in 2 bytes (run_trainer, bf16_state):
    the bf16 mode of the HIP path is compared on identical operands.  Stored: the six losses, |grad| checksums of every
    parameter, the first 4096 elements of the named gradients, the post-step SN / BN state."""
    st = run_trainer(R, adv_loss="hinge", ch=4, T=16, k=4, B=2, n_class=3, steps=1, seed=170, z_dim=120, lr=5e-5,
                     grads_of=F14_NAMES, grad_head=4096, bf16_state=True)
    keep = {k: v for k, v in st.items() if not (".sd1." in k and v.size >= 4096)}
    save("f14_default_init_bf16", keep)


F15_NAMES = ("conv.0.cells.0.update_gate.weight", "conv.0.cells.1.reset_gate.weight", "conv.3.cells.2.out_gate.weight",
             "conv.6.cells.1.update_gate.bias", "conv.9.cells.2.out_gate.weight", "conv.9.cells.0.reset_gate.weight",
             "conv.1.conv0.module.weight_bar", "conv.11.conv1.module.weight_bar", "conv.10.CBNorm2.embed.weight",
             "embedding.weight", "affine_transfrom.weight", "colorize.module.weight_bar",
             "pre_conv.0.module.weight_bar", "attn.gamma", "self_attn.gamma", "linear.module.weight_bar",
             "res3d.conv1.module.weight_bar", "conv1.conv_sc.module.weight_bar")


def f15_state_carry(R):
    """BASELINE configs[4] as a STEP: two steps of the reference Trainer with initial ConvGRU states supplied at the first
    frame of every ConvGRU (ConvGRU.py:104-118 through `conditional_generator_forward`), T=12 (D_t pools to T'=3), 128x128
    frames (latent_dim 8), ch=2, k=4, B=2, 3 classes, hinge, lr 2e-3.  Clips, large weights and the twelve states are
    synth.py closed forms; stored: draws, losses, named gradients, |grad| checksums, the state gradients (heads + checksums),
    post-step SN / BN state."""
    st = run_trainer(R, adv_loss="hinge", ch=2, T=12, k=4, B=2, n_class=3, steps=2, seed=182, z_dim=16, lr=2e-3,
                     grads_of=F15_NAMES, synth_big=True, grad_head=8192, latent_dim=8, hidden=True)
    keep = {k: v for k, v in st.items() if not (".sd1." in k and v.size >= 4096)}
    save("f15_state_carry", keep)


def f16_full_width_128(R):
    """ONE step of the reference Trainer at BASELINE configs[3]'s per-clip shape: ch=32, T=48, 128x128 (latent_dim 8),
    600 classes, k=8, hinge, lr 5e-5, B=1 -- the real 128..1024-channel widths on the 8x8 .. 128x128 grids.  Same storage
    scheme as F11."""
    st = run_trainer(R, adv_loss="hinge", ch=32, T=48, k=8, B=1, n_class=600, steps=1, seed=190, z_dim=120, lr=5e-5,
                     grads_of=F11_NAMES, synth_big=True, grad_head=8192, latent_dim=8)
    keep = {k: v for k, v in st.items() if not (".sd1." in k and v.size >= 4096)}
    save("f16_full_width_128", keep)


def f12_ucf101_reader(R):
    """Real-data input path: a tiny synthetic UCF-101-style JPEG folder (2 classes, 3 videos of 9-14 frames, 40x30 pixels)
    read through the reference's UCF101 dataset (Dataloader/datasets/ucf101.py) with the training transforms main.py:42-55
    builds (multi-scale corner / random crop -> 16x16, random flip, ToTensor(255), Normalize(.5,.5)), T = 8, under seeded
    `random`.  Stored: every file of the folder (bytes), the annotation JSON, and for a list of (index, seed, crop mode) the
    clip tensor [3,T,16,16] + label the reference returned."""
    import io, json, random, tempfile
    from PIL import Image
    sys.path.insert(0, REF)
    from Dataloader.datasets.ucf101 import UCF101
    from Dataloader.transform.spatial_transforms import (Compose, Normalize, MultiScaleCornerCrop, MultiScaleRandomCrop,
                                                         RandomHorizontalFlip, ToTensor)
    from Dataloader.transform.temporal_transforms import TemporalRandomCrop
    from Dataloader.transform.target_transforms import ClassLabel
    rng = np.random.RandomState(12)
    st = {}
    root = tempfile.mkdtemp()
    vids = [("ApplyEyeMakeup", "v_a_g01_c01", 14, "training"), ("Archery", "v_b_g01_c02", 9, "training"),
            ("Archery", "v_b_g02_c01", 5, "training"), ("Archery", "v_b_g03_c01", 11, "validation")]
    files = []
    for cls, vid, n, _ in vids:
        d = os.path.join(root, "jpg", cls, vid)
        os.makedirs(d)
        base = rng.rand(30, 40, 3)
        for i in range(1, n + 1):
            img = np.clip(255 * (0.6 * base + 0.4 * np.sin(np.linspace(0, 3, 40))[None, :, None] * (i / n) + 0.1 * rng.rand(30, 40, 3)), 0, 255)
            Image.fromarray(img.astype(np.uint8)).save(os.path.join(d, "image_{:05d}.jpg".format(i)), quality=90)
        open(os.path.join(d, "n_frames"), "w").write(str(n) + "\n")
    ann = {"labels": ["ApplyEyeMakeup", "Archery"],
           "database": {vid: {"subset": sub, "annotations": {"label": cls}} for cls, vid, n, sub in vids}}
    ann_path = os.path.join(root, "ucf101_01.json")
    json.dump(ann, open(ann_path, "w"))
    for dp, _, fns in os.walk(root):
        for fn in sorted(fns):
            rel = os.path.relpath(os.path.join(dp, fn), root)
            files.append(rel)
            st["file." + rel] = np.frombuffer(open(os.path.join(dp, fn), "rb").read(), dtype=np.uint8)
    st["meta.files"] = np.array(files)
    scales = [1.0]
    for _ in range(1, 5):
        scales.append(scales[-1] * 0.84089641525)
    cases = []
    for mode in ("corner", "random", "center"):
        crop = (MultiScaleRandomCrop(scales, 16) if mode == "random" else
                MultiScaleCornerCrop(scales, 16) if mode == "corner" else MultiScaleCornerCrop(scales, 16, crop_positions=["c"]))
        spatial = Compose([crop, RandomHorizontalFlip(), ToTensor(255), Normalize([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])])
        ds = UCF101(os.path.join(root, "jpg"), ann_path, "training", spatial_transform=spatial,
                    temporal_transform=TemporalRandomCrop(8), target_transform=ClassLabel())
        assert len(ds) == 3
        for index in range(3):
            for seed in (1, 2, 3):
                random.seed(100 * seed + index)
                clip, label = ds[index]
                tag = f"{mode}.{index}.{seed}"
                st["out.clip." + tag], st["out.label." + tag] = npy(clip), np.array(label)
                cases.append(tag)
    st["meta.cases"] = np.array(cases)
    save("f12_ucf101_reader", st)


ALL = {"f1": f1_spectral_norm, "f2": f2_conditional_norm, "f3": f3_gresblock, "f4": f4_convgru,
       "f5": f5_attention, "f6": f6_generator, "f7": f7_discriminators, "f8": f8_helpers,
       "f9": f9_trainer_steps, "f10": f10_config1, "f11": f11_full_width,
       "f12": f12_ucf101_reader, "f13": f13_two_d_iters, "f14": f14_default_init_bf16,
       "f15": f15_state_carry, "f16": f16_full_width_128}

if __name__ == "__main__":
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(False)
    sys.path.insert(0, HERE)
    R = import_reference()
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print(w)
        ALL[w](R)
