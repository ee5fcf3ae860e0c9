"""CPU oracle for the DVD-GAN G + D_s + D_t training path.   *** TEST INFRASTRUCTURE ***

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it.  The shipped path (dvd_gan_amd) never routes
through here and fails loudly when its HIP library is missing.

What it is: a from-scratch functional restatement, in fp32 torch CPU ops, of the algorithm the
reference implements in /root/reference (file:line cited per function).  State lives in flat
dicts keyed exactly like the reference `state_dict()`s, so reference checkpoints and the golden
vectors in tests/golden/ (generated from the real reference by tests/golden/make_golden.py)
plug straight in.  Parity is PINNED: tests/test_oracle_golden.py checks every function below
against those vectors (F1..F10 of SURVEY.md section 8c, F11..F16 added by later rounds).

All reference quirks are reproduced on purpose (SURVEY.md section 8a notes 1-6):
  * SN power iteration advances u/v on every forward, also in eval / no_grad;
  * the generator's condition rows are t-major while frames are b-major;
  * the generator loss under hinge is relu(1 - D(G(z)));
  * 'wgan-gp' is plain mean(+-x), the gradient penalty is dead code;
  * (found while pinning F9) SN backward reads the u / v of the LATEST forward, not of the
    forward that built the graph -- see sn_weight().
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# state helpers
# --------------------------------------------------------------------------------------
_NO_GRAD_SUFFIX = ("weight_u", "weight_v", "running_mean", "running_var", "num_batches_tracked")


def is_trainable(key):
    """u / v are Parameter(requires_grad=False) (Normalization.py:49-50) and BN buffers are
    buffers; everything else in the three state_dicts is trained (trainer.py:136-141)."""
    return not key.endswith(_NO_GRAD_SUFFIX)


def make_state(arrays, requires_grad=True, dtype=torch.float32):
    """dict of numpy / tensors (reference state_dict keys) -> dict of leaf tensors (fp32; fp64 to measure what fp32 rounding
    alone does to a fixture)."""
    sd = {}
    for k, v in arrays.items():
        t = torch.as_tensor(v).clone()
        if t.is_floating_point():
            t = t.to(dtype)
            if requires_grad and is_trainable(k):
                t.requires_grad_(True)
        sd[k] = t
    return sd


def trainable(sd):
    return {k: v for k, v in sd.items() if v.is_floating_point() and is_trainable(k)}


# --------------------------------------------------------------------------------------
# SpectralNorm / ConditionalNorm                                  Module/Normalization.py
# --------------------------------------------------------------------------------------
def l2normalize(v, eps=1e-12):
    """Normalization.py:7-8 -- eps is added to the norm, not clamped."""
    return v / (v.norm() + eps)


def sn_weight(sd, pfx):
    """SpectralNorm._update_u_v, Normalization.py:19-31.  One power iteration on the
    [out, rest] matrix view, in-place update of `weight_u` / `weight_v` (every call),
    sigma = u . (W v) differentiable through W only, returns W_bar / sigma."""
    u, v, w = sd[pfx + "weight_u"], sd[pfx + "weight_v"], sd[pfx + "weight_bar"]
    mat = w.reshape(w.shape[0], -1)
    with torch.no_grad():
        v_new = l2normalize(mat.t().mv(u))
        u_new = l2normalize(mat.mv(v_new))
    # The reference REBINDS `.data` (Normalization.py:26-27): no version bump, and autograd
    # graphs built by an EARLIER forward keep pointing at the same u / v objects.  When a
    # module runs twice before one backward (D on real then on fake, trainer.py:243-251) the
    # first pass's d(sigma)/dW is therefore evaluated with the SECOND pass's u, v (quirk 7,
    # pinned by golden F9).  Rebinding here reproduces that exactly.
    v.data = v_new
    u.data = u_new
    sigma = u.dot(mat.mv(v))
    return w / sigma


def conditional_norm(sd, pfx, x, cond, training=True):
    """ConditionalNorm.forward, Normalization.py:78-88: BatchNorm2d(affine=False, eps=1e-5,
    momentum=0.1) then gamma, beta = Linear(cond).chunk(2)."""
    rm, rv = sd[pfx + "bn.running_mean"], sd[pfx + "bn.running_var"]
    if training:
        sd[pfx + "bn.num_batches_tracked"] += 1
    out = F.batch_norm(x, rm, rv, None, None, training, 0.1, 1e-5)
    emb = F.linear(cond, sd[pfx + "embed.weight"], sd[pfx + "embed.bias"])
    c = x.shape[1]
    gamma, beta = emb[:, :c], emb[:, c:]
    return gamma.reshape(-1, c, 1, 1) * out + beta.reshape(-1, c, 1, 1)


# --------------------------------------------------------------------------------------
# ConvGRU                                                              Module/ConvGRU.py
# --------------------------------------------------------------------------------------
def convgru_cell(sd, pfx, x, h):
    """ConvGRUCell.forward, ConvGRU.py:29-54.  h=None -> zeros."""
    wu, bu = sd[pfx + "update_gate.weight"], sd[pfx + "update_gate.bias"]
    wr, br = sd[pfx + "reset_gate.weight"], sd[pfx + "reset_gate.bias"]
    wo, bo = sd[pfx + "out_gate.weight"], sd[pfx + "out_gate.bias"]
    pad = wu.shape[-1] // 2
    if h is None:
        h = x.new_zeros(x.shape[0], wu.shape[0], x.shape[2], x.shape[3])
    s = torch.cat([x, h], 1)
    u = torch.sigmoid(F.conv2d(s, wu, bu, padding=pad))
    r = torch.sigmoid(F.conv2d(s, wr, br, padding=pad))
    o = torch.tanh(F.conv2d(torch.cat([x, h * r], 1), wo, bo, padding=pad))
    return h * (1 - u) + o * u


def convgru(sd, pfx, x, hidden, n_layers=3):
    """ConvGRU.forward, ConvGRU.py:104-133: layer l consumes layer l-1's NEW state."""
    if hidden is None:
        hidden = [None] * n_layers
    out, inp = [], x
    for l in range(n_layers):
        inp = convgru_cell(sd, f"{pfx}cells.{l}.", inp, hidden[l])
        out.append(inp)
    return out


# --------------------------------------------------------------------------------------
# GResBlock                                                          Module/GResBlock.py
# --------------------------------------------------------------------------------------
def gresblock(sd, pfx, x, cond, upsample, training=True):
    """GResBlock.forward, GResBlock.py:42-86 (bn=True, downsample_factor=1, skip always
    projected by the 1x1 SN conv)."""
    out = F.relu(conditional_norm(sd, pfx + "CBNorm1.", x, cond, training))
    if upsample != 1:
        out = F.interpolate(out, scale_factor=upsample)
    out = F.conv2d(out, sn_weight(sd, pfx + "conv0.module."), sd[pfx + "conv0.module.bias"], padding=1)
    out = F.relu(conditional_norm(sd, pfx + "CBNorm2.", out, cond, training))
    out = F.conv2d(out, sn_weight(sd, pfx + "conv1.module."), sd[pfx + "conv1.module.bias"], padding=1)
    skip = x
    if upsample != 1:
        skip = F.interpolate(skip, scale_factor=upsample)
    skip = F.conv2d(skip, sn_weight(sd, pfx + "conv_sc.module."), sd[pfx + "conv_sc.module.bias"])
    return out + skip


# --------------------------------------------------------------------------------------
# Generator                                                          Module/Generator.py
# --------------------------------------------------------------------------------------
GEN_STACK = ("gru", "res1", "res2") * 4      # Generator.py:38-55: [ConvGRU, GResBlock, GResBlock(up)] x 4


def generator(sd, z, class_id, ch, n_frames, latent_dim=4, training=True, taps=None, hidden=None, self_attn=False,
              sep_attn=False):
    """Generator.forward, Generator.py:63-120 (hierar_flag=False).
    taps (test aid): a list that receives the input of the first module followed by the output of each of the 12
    modules of `self.conv` ([B*T, C, S, S], b-major frames), each with retain_grad() so a later backward leaves the
    gradient flowing through that point on the tensor.
    hidden (frame-conditional variant): per ConvGRU of the stack (4 entries), None or the list of per-layer states that
    replaces the `hidden=None` of the first frame (Generator.py:91,96 -> ConvGRU.forward(x, hidden), ConvGRU.py:104-118).
    self_attn / sep_attn: the two attention blocks the reference defines and imports (Generator.py:10) but leaves commented
    out (:29 `self.self_attn = SelfAttention(8 * ch)`, :34 `SeparableAttn(4 * ch)` after the block that produces 4*ch
    channels): SelfAttention over the (T, 4, 4) latent clip after the first ConvGRU, SeparableAttn over the (T, 32, 32) clip
    after module 8; state keys `self_attn.*` / `sep_attn.model.{0,1,2}.*`."""
    B, T = z.shape[0], n_frames
    class_emb = F.embedding(class_id, sd["embedding.weight"])
    zc = torch.cat([z, class_emb], 1)
    y = F.linear(zc, sd["affine_transfrom.weight"], sd["affine_transfrom.bias"])
    y = y.view(-1, 8 * ch, latent_dim, latent_dim)
    cond = zc.repeat(T, 1)                    # Generator.py:109-110: t-major rows (quirk 1)

    def tap(v):
        if taps is not None:
            if v.requires_grad:
                v.retain_grad()
            taps.append(v)
    tap(y)
    for k, kind in enumerate(GEN_STACK):
        pfx = f"conv.{k}."
        if kind == "gru":
            if k > 0:
                y = y.view(B, T, *y.shape[1:])
            state, frames = (hidden[k // 3] if hidden is not None else None), []
            for t in range(T):
                state = convgru(sd, pfx, y if k == 0 else y[:, t], state)
                frames.append(state[-1])
            y = torch.stack(frames, 1).reshape(B * T, *frames[0].shape[1:])    # b-major frames
        else:
            y = gresblock(sd, pfx, y, cond, 1 if kind == "res1" else 2, training)
        if (self_attn and k == 0) or (sep_attn and k == 8):
            clip = y.view(B, T, *y.shape[1:]).permute(0, 2, 1, 3, 4)           # [B, C, T, W, H]
            clip = self_attention_3d(sd, "self_attn.", clip) if k == 0 else separable_attn(sd, "sep_attn.", clip)
            y = clip.permute(0, 2, 1, 3, 4).reshape(B * T, *y.shape[1:])
        tap(y)
    y = F.relu(y)
    y = F.conv2d(y, sn_weight(sd, "colorize.module."), sd["colorize.module.bias"], padding=1)
    y = torch.tanh(y)
    return y.view(B, T, *y.shape[1:])


# --------------------------------------------------------------------------------------
# Discriminators                                                Module/Discriminators.py
# --------------------------------------------------------------------------------------
def self_attention_2d(sd, pfx, x):
    """SelfAttention.forward, Discriminators.py:100-119: softmax(Q^T K) with no 1/sqrt(d),
    out = gamma * V A^T + x."""
    B, C, W, H = x.shape
    n = W * H
    q = F.conv2d(x, sd[pfx + "query_conv.weight"], sd[pfx + "query_conv.bias"]).view(B, -1, n)
    k = F.conv2d(x, sd[pfx + "key_conv.weight"], sd[pfx + "key_conv.bias"]).view(B, -1, n)
    v = F.conv2d(x, sd[pfx + "value_conv.weight"], sd[pfx + "value_conv.bias"]).view(B, -1, n)
    att = torch.softmax(torch.bmm(q.transpose(1, 2), k), -1)
    out = torch.bmm(v, att.transpose(1, 2)).view(B, C, W, H)
    return sd[pfx + "gamma"] * out + x


def _dblock(sd, pfx, x, conv, pool):
    """GBlock / Res3dBlock with bn=False, upsample=False, downsample=True
    (Discriminators.py:180-211 / 335-366)."""
    out = F.relu(x)
    out = conv(out, sn_weight(sd, pfx + "conv0.module."), sd[pfx + "conv0.module.bias"], padding=1)
    out = F.relu(out)
    out = conv(out, sn_weight(sd, pfx + "conv1.module."), sd[pfx + "conv1.module.bias"], padding=1)
    out = pool(out, 2)
    skip = conv(x, sn_weight(sd, pfx + "conv_sc.module."), sd[pfx + "conv_sc.module.bias"])
    return out + pool(skip, 2)


def gblock(sd, pfx, x):
    return _dblock(sd, pfx, x, F.conv2d, F.avg_pool2d)


def res3d_block(sd, pfx, x):
    return _dblock(sd, pfx, x, F.conv3d, F.avg_pool3d)


def _proj_head(sd, feat, class_id, repeat):
    """Shared head, Discriminators.py:264-291 / 421-447: relu -> sum(H,W) -> SN linear +
    sum_c h_c * SN-embed[class]_c, class ids repeated per frame b-major."""
    out = F.relu(feat)
    out = out.view(out.shape[0], out.shape[1], -1).sum(2)
    lin = F.linear(out, sn_weight(sd, "linear.module."), sd["linear.module.bias"]).squeeze(1)
    cid = class_id.view(-1, 1).repeat(1, repeat).view(-1)
    emb = F.embedding(cid, sn_weight(sd, "embed.module."))
    return lin + (out * emb).sum(1)


def spatial_disc(sd, x, class_id):
    """SpatialDiscriminator.forward, Discriminators.py:242-291.  x: [B,k,3,H,W]."""
    B, T, C, W, H = x.shape
    x = x.reshape(B * T, C, H, W)
    out = F.conv2d(x, sn_weight(sd, "pre_conv.0.module."), sd["pre_conv.0.module.bias"], padding=1)
    out = F.relu(out)
    out = F.conv2d(out, sn_weight(sd, "pre_conv.2.module."), sd["pre_conv.2.module.bias"], padding=1)
    out = F.avg_pool2d(out, 2)
    out = out + F.conv2d(F.avg_pool2d(x, 2), sn_weight(sd, "pre_skip.module."), sd["pre_skip.module.bias"])
    out = gblock(sd, "conv1.", out)
    out = self_attention_2d(sd, "attn.", out)
    for i in range(3):
        out = gblock(sd, f"conv2.{i}.", out)
    return _proj_head(sd, out, class_id, T)


def temporal_disc(sd, x, class_id):
    """TemporalDiscriminator.forward, Discriminators.py:400-447.  x: [B,3,T,h,w]."""
    out = F.conv3d(x, sn_weight(sd, "pre_conv.0.module."), sd["pre_conv.0.module.bias"], padding=1)
    out = F.relu(out)
    out = F.conv3d(out, sn_weight(sd, "pre_conv.2.module."), sd["pre_conv.2.module.bias"], padding=1)
    out = F.avg_pool3d(out, 2)
    out = out + F.conv3d(F.avg_pool3d(x, 2), sn_weight(sd, "pre_skip.module."), sd["pre_skip.module.bias"])
    out = res3d_block(sd, "res3d.", out)
    out = out.permute(0, 2, 1, 3, 4).contiguous()
    B, T, C, W, H = out.shape
    out = out.view(B * T, C, W, H)
    out = self_attention_2d(sd, "self_attn.", out)
    for i in range(3):
        out = gblock(sd, f"conv.{i}.", out)
    return _proj_head(sd, out, class_id, T)


# --------------------------------------------------------------------------------------
# helpers                                                                         utils.py
# --------------------------------------------------------------------------------------
def frame_ids_from_perm(perm, k_sample):
    """utils.py:61-62: first k entries of a permutation, sorted (k > T -> all frames)."""
    return torch.as_tensor(perm)[:k_sample].sort()[0]


def sample_k_frames(data, frame_ids):
    """utils.py:60-63 with the permutation supplied by the caller: the SAME ids for the batch."""
    return data[:, frame_ids]


def vid_downsample(data):
    """utils.py:77-83: per-frame 2x2 average pool, [B,T,C,H,W] -> [B,C,T,H/2,W/2]."""
    B, T, C, H, W = data.shape
    x = F.avg_pool2d(data.reshape(B * T, C, H, W), 2)
    return x.view(B, T, C, H // 2, W // 2).permute(0, 2, 1, 3, 4).contiguous()


def adv_loss(x, real_flag, kind):
    """Trainer.calc_loss, trainer.py:114-121."""
    if real_flag:
        x = -x
    if kind == "wgan-gp":
        return x.mean()
    return F.relu(1.0 + x).mean()


# --------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: eps 1e-8, no weight decay, no amsgrad)  trainer.py:136-141
# --------------------------------------------------------------------------------------
class Adam:
    def __init__(self, params, lr, betas=(0.0, 0.9), eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = params, lr, betas[0], betas[1], eps
        self.t = 0
        self.m = {k: torch.zeros_like(p) for k, p in params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in params.items()}

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2_sqrt = math.sqrt(1 - self.b2 ** self.t)
        for k, p in self.params.items():
            if p.grad is None:
                continue
            g = p.grad
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


# --------------------------------------------------------------------------------------
# one training step                                                    trainer.py:213-307
# --------------------------------------------------------------------------------------
class TrainState:
    """G / D_s / D_t states + three Adam instances (trainer.py:134-141, 345-351)."""

    def __init__(self, g_sd, ds_sd, dt_sd, *, ch, n_frames, k_sample, n_class, z_dim=120,
                 latent_dim=4, adv="hinge", g_lr=5e-5, d_lr=5e-5, betas=(0.0, 0.9)):
        self.G, self.Ds, self.Dt = g_sd, ds_sd, dt_sd
        self.ch, self.T, self.k, self.n_class, self.z_dim = ch, n_frames, k_sample, n_class, z_dim
        self.latent_dim, self.adv = latent_dim, adv
        self.g_opt = Adam(trainable(g_sd), g_lr, betas)
        self.ds_opt = Adam(trainable(ds_sd), d_lr, betas)
        self.dt_opt = Adam(trainable(dt_sd), d_lr, betas)

    def zero_grad(self):
        """Trainer.reset_grad, trainer.py:384-387 (all three optimizers)."""
        for opt in (self.ds_opt, self.dt_opt, self.g_opt):
            for p in opt.params.values():
                p.grad = None


def snapshot_grads(st):
    """Test aid: record every parameter gradient at the moment its optimizer steps (the gradients the reference's
    three `optimizer.step()` calls consume, trainer.py:253,269,307).  Returns {"Ds" | "Dt" | "G": {key: grad}}, filled
    by the next train_step."""
    snaps = {}
    for tag, opt in (("Ds", st.ds_opt), ("Dt", st.dt_opt), ("G", st.g_opt)):
        def stepper(opt=opt, tag=tag, orig=opt.step):
            snaps[tag] = {k: p.grad.detach().clone() for k, p in opt.params.items() if p.grad is not None}
            orig()
        opt.step = stepper
    return snaps


def train_step(st, real_videos, real_labels, z, z_class, perm_real, perm_fake, rec=None, d_iters=1, hidden=None):
    """trainer.py:213-307.  real_videos: [B,3,T,H,W]; RNG draws are passed in the order the reference consumes them:
    perm_real, z, z_class, perm_fake -- for d_iters > 1 (the loop of trainer.py:230) each of the four is a sequence with one
    entry per discriminator iteration; the generator step uses the clips of the LAST iteration (trainer.py:296-297).
    Returns the six loss terms [ds_real, ds_fake, dt_real, dt_fake, g_s, g_t] (discriminator terms of the last iteration).
    rec (test aid): a dict that receives the generator taps (`generator(..., taps=)`), the generated clips and the six
    raw discriminator outputs, for teacher-forced comparisons of single modules.
    hidden (frame-conditional variant, BASELINE configs[4]): initial ConvGRU states handed to `generator(hidden=)` in every
    generator forward of the step (ConvGRU.py:104-118); leaves with requires_grad receive their gradient from the generator
    update's backward pass (the discriminator updates see detached clips).  Pinned by golden F15."""
    taps = [] if rec is not None else None
    real = real_videos.permute(0, 2, 1, 3, 4).contiguous()                      # :227
    if d_iters == 1:
        z, z_class, perm_real, perm_fake = [z], [z_class], [perm_real], [perm_fake]
    for it in range(d_iters):                                                   # :230
        if taps is not None:
            del taps[:]
        real_s = sample_k_frames(real, frame_ids_from_perm(perm_real[it], st.k))       # :233
        zc = z_class[it]
        fake = generator(st.G, z[it], zc, st.ch, st.T, st.latent_dim, taps=taps, hidden=hidden)      # :239
        fake_s = sample_k_frames(fake, frame_ids_from_perm(perm_fake[it], st.k))       # :242
        # ---- D_s ----                                                             :243-253
        o_sr, o_sf = spatial_disc(st.Ds, real_s, real_labels), spatial_disc(st.Ds, fake_s.detach(), zc)
        ds_real = adv_loss(o_sr, True, st.adv)
        ds_fake = adv_loss(o_sf, False, st.adv)
        st.zero_grad()
        (ds_real + ds_fake).backward()
        st.ds_opt.step()
        # ---- D_t ----                                                             :256-269
        real_d, fake_d = vid_downsample(real), vid_downsample(fake)
        o_tr, o_tf = temporal_disc(st.Dt, real_d, real_labels), temporal_disc(st.Dt, fake_d.detach(), zc)
        dt_real = adv_loss(o_tr, True, st.adv)
        dt_fake = adv_loss(o_tf, False, st.adv)
        st.zero_grad()
        (dt_real + dt_fake).backward()
        st.dt_opt.step()
    z_class = zc
    # ---- G (on the UPDATED discriminators; loss uses the "real" form) ----        :296-307
    o_gs, o_gt = spatial_disc(st.Ds, fake_s, z_class), temporal_disc(st.Dt, fake_d, z_class)
    g_s = adv_loss(o_gs, True, st.adv)
    g_t = adv_loss(o_gt, True, st.adv)
    if rec is not None:
        fake.retain_grad()
        rec.update(taps=taps, fake=fake, real_s=real_s, fake_s=fake_s, real_d=real_d, fake_d=fake_d,
                   d_out={"ds_real": o_sr, "ds_fake": o_sf, "dt_real": o_tr, "dt_fake": o_tf, "g_s": o_gs, "g_t": o_gt})
    st.zero_grad()
    (g_s + g_t).backward()
    st.g_opt.step()
    return [float(v.detach()) for v in (ds_real, ds_fake, dt_real, dt_fake, g_s, g_t)]


# --------------------------------------------------------------------------------------
# (defined, never invoked by the reference)                           Module/Attention.py
# --------------------------------------------------------------------------------------
def self_attention_3d(sd, pfx, x):
    """Attention.SelfAttention.forward, Attention.py:153-185: T*H*W tokens, K and V
    2x2x2 max-pooled, q/k channels C/2."""
    B, C, T, W, H = x.shape
    n = T * W * H
    q = F.conv3d(x, sd[pfx + "query_conv.weight"], sd[pfx + "query_conv.bias"]).view(B, -1, n)
    k = F.max_pool3d(F.conv3d(x, sd[pfx + "key_conv.weight"], sd[pfx + "key_conv.bias"]), 2, 2)
    v = F.max_pool3d(F.conv3d(x, sd[pfx + "value_conv.weight"], sd[pfx + "value_conv.bias"]), 2, 2)
    k, v = k.view(B, -1, n // 8), v.view(B, -1, n // 8)
    att = torch.softmax(torch.bmm(q.transpose(1, 2), k), -1)
    out = torch.bmm(v, att.transpose(1, 2)).view(B, C, T, W, H)
    return sd[pfx + "gamma"] * out + x


def separable_attn_cell(sd, pfx, x, axis):
    """SeparableAttnCell.forward, Attention.py:61-111, for attn_id = axis in 'T' | 'W' | 'H'.  x: [B, C, T, W, H].
    The reference builds its "attention over one axis" from RAW RESHAPES of contiguous tensors (`.view(B, A, -1)` of a
    [B, C/2, A, ., .] tensor is not a transpose): restated literally --
        o  = x with the attended axis swapped into position 2 (T: x itself, W: transpose(2,3), H: transpose(2,4))
        q  = conv1x1(o)                       -> contiguous [B, C/2, A, d1, d2], REINTERPRETED as [B, A, L]
        k  = maxpool_(2,1,1)(conv1x1(o))      -> [B, C/2, A/2, d1, d2]          REINTERPRETED as [B, L, A/2]
        v  = maxpool_(2,1,1)(conv1x1(o))      -> [B, C, A/2, d1, d2]            REINTERPRETED as [B, R, A/2]
        out = v @ softmax(q @ k)^T            -> [B, R, A], reinterpreted as [B, C, (the other two axes), A] and permuted
        y  = gamma * out + x"""
    B, C, T, W, H = x.shape
    assert T % 2 == 0 and W % 2 == 0 and H % 2 == 0, "T, W, H is not even"
    if axis == "T":
        A, o = T, x
    elif axis == "W":
        A, o = W, x.transpose(2, 3)
    else:
        A, o = H, x.transpose(2, 4)
    q = F.conv3d(o, sd[pfx + "query_conv.weight"], sd[pfx + "query_conv.bias"]).contiguous().view(B, A, -1)
    k = F.max_pool3d(F.conv3d(o, sd[pfx + "key_conv.weight"], sd[pfx + "key_conv.bias"]), (2, 1, 1), (2, 1, 1))
    k = k.contiguous().view(B, -1, A // 2)
    att = torch.softmax(torch.bmm(q, k), -1)
    v = F.max_pool3d(F.conv3d(o, sd[pfx + "value_conv.weight"], sd[pfx + "value_conv.bias"]), (2, 1, 1), (2, 1, 1))
    v = v.contiguous().view(B, -1, A // 2)
    out = torch.bmm(v, att.transpose(2, 1))
    if axis == "T":
        out = out.view(B, C, W, H, T).permute(0, 1, 4, 2, 3)
    elif axis == "W":
        out = out.view(B, C, T, H, W).permute(0, 1, 2, 4, 3)
    else:
        out = out.view(B, C, T, W, H)
    return sd[pfx + "gamma"] * out + x


def separable_attn(sd, pfx, x):
    """SeparableAttn.forward, Attention.py:8-21: the T, W and H cells in sequence (keys `model.{0,1,2}.*`)."""
    for i, axis in enumerate("TWH"):
        x = separable_attn_cell(sd, f"{pfx}model.{i}.", x, axis)
    return x
