"""Spatial / temporal discriminators of DVD-GAN on the HIP kernels -- constructor / forward
signatures, state_dict keys and results of Module/Discriminators.py:217-291 (D_s), :369-447 (D_t),
GBlock :151-211, Res3dBlock :305-366 and the 2-D SelfAttention :82-119.

Every block here is the `bn=False, upsample=False, downsample=True` configuration the reference
instantiates.  Fusions: ReLU on the block input is applied while the conv kernel loads its tile,
ReLU between the two convs is the first conv's epilogue, the skip branch is added in the second
conv's epilogue so ONE average pool serves both branches (pool is linear), q/k/v are one GEMM.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import kern as K
from . import lib as L
from .sn_layers import PlainConv, SpectralNormConv, _SNInner, clear_spectral_norm, prefetch_spectral_norm


class SelfAttention(nn.Module):
    """Discriminators.py:82-119.  Keys: gamma, {query,key,value}_conv.{weight,bias}."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = PlainConv(in_dim, in_dim // 8, (1, 1), "xavier")
        self.key_conv = PlainConv(in_dim, in_dim // 8, (1, 1), "xavier")
        self.value_conv = PlainConv(in_dim, in_dim, (1, 1), "xavier")
        self.gamma = nn.Parameter(torch.zeros(1))
        self.train_weights = True

    def forward(self, x):
        C_, dq = self.chanel_in, self.chanel_in // 8
        dqp = K.pad8(dq)
        ctot = 2 * dqp + C_                               # q | pad | k | pad | v
        spec = Fn.ConvSpec((1, 1), ctot, C_)
        pk = K.PackedConv(x.dtype, ctot, C_, (1, 1), x.device)
        det = (lambda t: t) if self.train_weights else (lambda t: t.detach())
        wq, wk, wv = det(self.query_conv.weight), det(self.key_conv.weight), det(self.value_conv.weight)
        pk.fill(wq.data, co_off=0).fill(wk.data, co_off=dqp).fill(wv.data, co_off=2 * dqp)
        spec.pack = pk
        qkv = QKVConv.apply(x, wq, det(self.query_conv.bias), wk, det(self.key_conv.bias), wv,
                            det(self.value_conv.bias), spec, dq, dqp)
        return Fn.SelfAttention2d.apply(x, qkv, det(self.gamma), dq, C_)


class QKVConv(torch.autograd.Function):
    """The three 1x1 convolutions of the attention block as one GEMM with a fused output
    [q | k | v] (each sub-block starts on a multiple of 8 channels)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, spec, dq, dqp):
        ctot = spec.cout
        bias = torch.zeros(ctot, dtype=torch.float32, device=x.device)
        bias[:dq] = bq
        bias[dqp:dqp + dq] = bk
        bias[2 * dqp:] = bv
        qkv = K.conv_forward(x, spec.pack.wf, (1, 1), ctot, bias=bias)
        ctx.save_for_backward(x, wq, wk, wv)
        ctx.spec, ctx.dq, ctx.dqp = spec, dq, dqp
        ctx.params = (wq, bq, wk, bk, wv, bv)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        x, wq, wk, wv = ctx.saved_tensors
        wq_, bq_, wk_, bk_, wv_, bv_ = ctx.params
        spec, dq, dqp = ctx.spec, ctx.dq, ctx.dqp
        dqkv = dqkv.contiguous()
        C_ = spec.cin
        dx = K.conv_forward(dqkv, spec.pack.wd, (1, 1), spec.pack.cip) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1]
        parts = ((wq_, bq_, 0, dq), (wk_, bk_, dqp, dq), (wv_, bv_, 2 * dqp, C_))
        if need_w and Fn._direct(wq_, bq_, wk_, bk_, wv_, bv_):
            # persistent .grad buffers: weight and bias gradients are accumulated straight into them
            for wp, bp, off, n in parts:
                K.conv_wgrad(x, dqkv, wp.grad, (1, 1), n, C_, dy_col=off, dbias=bp.grad)
            return dx, None, None, None, None, None, None, None, None, None
        outs = []
        db = torch.zeros(spec.cout, dtype=torch.float32, device=x.device) if need_w else None
        for w, _, off, n in parts:
            if need_w:
                dw = torch.zeros_like(w)
                K.conv_wgrad(x, dqkv, dw, (1, 1), n, C_, dy_col=off, dbias=db[off:off + n])
                outs.append(dw)
            else:
                outs.append(None)
        dbq = db[:dq].clone() if need_w else None
        dbk = db[dqp:dqp + dq].clone() if need_w else None
        dbv = db[2 * dqp:].clone() if need_w else None
        return dx, outs[0], dbq, outs[1], dbk, outs[2], dbv, None, None, None


class GBlock(nn.Module):
    """Discriminators.py:151-211 (bn=False, upsample=False, downsample=True) and, with 3-D kernels
    and (2,2,2) pooling, Res3dBlock :305-366."""

    def __init__(self, in_channel, out_channel, three_d=False):
        super().__init__()
        k = (3, 3, 3) if three_d else (3, 3)
        k1 = (1, 1, 1) if three_d else (1, 1)
        self.pt = 2 if three_d else 1
        self.conv0 = SpectralNormConv(in_channel, out_channel, k)
        self.conv1 = SpectralNormConv(out_channel, out_channel, k)
        self.conv_sc = SpectralNormConv(in_channel, out_channel, k1)

    def forward(self, x):
        a1 = self.conv0(x, relu_in=True, act=L.ACT_RELU)
        skip = self.conv_sc(x)
        return Fn.Pool.apply(self.conv1(a1, res=skip), self.pt)


class Res3dBlock(GBlock):
    def __init__(self, in_channel, out_channel):
        super().__init__(in_channel, out_channel, three_d=True)


class _SNHolder(nn.Module):
    """SpectralNorm(nn.Linear) / SpectralNorm(nn.Embedding): only the `.module.*` parameters."""

    def __init__(self, shape, bias_n, init_weight=None):
        super().__init__()
        self.module = _SNInner(shape, bias_n, init_weight)


def _check_frame_size(H, W):
    """Any even frame size runs (Discriminators.py:242-291, 400-447 take what their poolings leave): power-of-two extents
    through the LDS-staged kernels, any other extent (96 x 96, 80 x 80, ...) through the tap-by-tap kernels with division
    indexing; maps that become odd further down (3 x 3 -> 1 x 1) are floored by the pooling like F.avg_pool2d.  An odd INPUT
    frame would lose a line in the very first pooling -- the reference floors silently, here it is an error."""
    if H < 2 or W < 2 or (H | W) & 1:
        raise ValueError(f"{H}x{W} frames: the discriminators need even frame sizes")


class _DiscBase(nn.Module):
    def _set_train_weights(self, flag):
        for m in self.modules():
            if hasattr(m, "train_weights"):
                m.train_weights = flag

    def _head(self, feat, class_id, repeat):
        C_ = self.linear.module.weight_bar.shape[1]
        cls = class_id.view(-1, 1).repeat(1, repeat).view(-1).to(torch.int32)       # b-major frames
        lin, emb = self.linear.module, self.embed.module
        det = (lambda t: t) if self.train_weights else (lambda t: t.detach())
        return Fn.ProjectionHead.apply(feat, det(lin.weight_bar), det(lin.bias), det(emb.weight_bar), cls,
                                       (lin.weight_u.data, lin.weight_v.data), (emb.weight_u.data, emb.weight_v.data), C_)

    def _make_head(self, chn, n_class):
        self.linear = _SNHolder((1, 16 * chn), 1)
        self.embed = _SNHolder((n_class, 16 * chn), 0, lambda w: w.uniform_(-0.1, 0.1))
        self.train_weights = True


class SpatialDiscriminator(_DiscBase):
    """SpatialDiscriminator(chn=128, n_class=4).forward(x [B,k,3,H,W], class_id [B]) -> [B*k]"""

    def __init__(self, chn=128, n_class=4, compute_dtype=torch.bfloat16):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.pre_conv = nn.ModuleList([SpectralNormConv(3, 2 * chn, (3, 3)), nn.Identity(),
                                       SpectralNormConv(2 * chn, 2 * chn, (3, 3)), nn.Identity()])
        self.pre_skip = SpectralNormConv(3, 2 * chn, (1, 1))
        self.conv1 = GBlock(2 * chn, 4 * chn)
        self.attn = SelfAttention(4 * chn)
        self.conv2 = nn.Sequential(GBlock(4 * chn, 8 * chn), GBlock(8 * chn, 16 * chn), GBlock(16 * chn, 16 * chn))
        self._make_head(chn, n_class)

    def forward(self, x, class_id):
        sn = prefetch_spectral_norm(self, self.compute_dtype)
        try:
            return self._forward(x, class_id)
        finally:
            clear_spectral_norm(sn)

    def _forward(self, x, class_id):
        B, T, C_, H, W = x.shape
        _check_frame_size(H, W)
        xc = Fn.ToChannelsLast.apply(x.reshape(B * T, C_, H, W), self.compute_dtype, None)
        c1 = self.pre_conv[0](xc, act=L.ACT_RELU)
        p2 = Fn.Pool.apply(self.pre_conv[2](c1), 1)
        out = self.pre_skip(Fn.Pool.apply(xc, 1), res=p2)
        out = self.conv1(out)
        out = self.attn(out)
        out = self.conv2(out)
        return self._head(out, class_id, T)


class TemporalDiscriminator(_DiscBase):
    """TemporalDiscriminator(chn=128, n_class=4).forward(x [B,3,T,h,w], class_id [B]) -> [B*T/4]"""

    def __init__(self, chn=128, n_class=4, compute_dtype=torch.bfloat16):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.pre_conv = nn.ModuleList([SpectralNormConv(3, 2 * chn, (3, 3, 3)), nn.Identity(),
                                       SpectralNormConv(2 * chn, 2 * chn, (3, 3, 3)), nn.Identity()])
        self.pre_skip = SpectralNormConv(3, 2 * chn, (1, 1, 1))
        self.res3d = Res3dBlock(2 * chn, 4 * chn)
        self.self_attn = SelfAttention(4 * chn)
        self.conv = nn.Sequential(GBlock(4 * chn, 8 * chn), GBlock(8 * chn, 16 * chn), GBlock(16 * chn, 16 * chn))
        self._make_head(chn, n_class)

    def forward(self, x, class_id):
        sn = prefetch_spectral_norm(self, self.compute_dtype)
        try:
            return self._forward(x, class_id)
        finally:
            clear_spectral_norm(sn)

    def _forward(self, x, class_id):
        _check_frame_size(x.shape[-2], x.shape[-1])
        if x.shape[2] % 4:
            # two (2,2,2) average pools: the reference floors an odd length silently (Discriminators.py:380,408); the
            # pooled-gradient kernels here need whole windows, so say it up front instead of failing in backward
            raise ValueError(f"TemporalDiscriminator: n_frames={x.shape[2]} must be a multiple of 4 (two temporal 2x poolings)")
        xc = Fn.ToChannelsLast.apply(x, self.compute_dtype, None)            # [B,T,h,w,8]
        c1 = self.pre_conv[0](xc, act=L.ACT_RELU)
        p2 = Fn.Pool.apply(self.pre_conv[2](c1), 2)
        out = self.pre_skip(Fn.Pool.apply(xc, 2), res=p2)
        out = self.res3d(out)                                                # [B,T/4,h/4,w/4,4chn]
        B, T4 = out.shape[0], out.shape[1]
        if out.shape[2] < 8 or out.shape[3] < 8:
            # three more 2x2 pools follow: the reference raises here as well (quirk 4)
            raise RuntimeError("TemporalDiscriminator needs frames of at least 64x64 (32x32 after vid_downsample)")
        out = out.reshape(B * T4, out.shape[2], out.shape[3], out.shape[4])  # frames b-major, no permute needed
        out = self.self_attn(out)
        out = self.conv(out)
        return self._head(out, class_id, T4)
