"""Tensor-level wrappers over the C ABI (one Python function per entry point).

Internal activation layout: channels-last, [frames, (T,) H, W, Cp] with Cp = channels padded to a
multiple of 8, storage dtype torch.float32 (exact mode) or torch.bfloat16.
"""
import ctypes as C

import torch

from . import lib as L


def pad8(c):
    return (c + 7) // 8 * 8


def _ksize3(k):
    k = tuple(k)
    return (1,) * (3 - len(k)) + k


# ------------------------------------------------------------------ layout
def to_cl(x, dtype, swap=None):
    """fp32 [F, C, *spatial] -> [F, *spatial, pad8(C)] in `dtype`.  swap=(A, B): the F = A*B source
    frames (a-major) are written b-major."""
    x = x.contiguous()
    F_, Cc = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    P = 1
    for s in sp:
        P *= s
    out = torch.empty((F_,) + sp + (pad8(Cc),), dtype=dtype, device=x.device)
    A, B = swap if swap else (F_, 1)
    L.check(L.lib().dvd_to_channels_last(L.dt(out), L.ptr(x), L.ptr(out), C.c_longlong(F_), Cc, C.c_longlong(P),
                                         pad8(Cc), A, B, 1 if swap else 0, L.stream()))
    return out


def from_cl(t, channels, swap=None):
    """[F, *spatial, Cp] -> fp32 [F, channels, *spatial] (inverse of to_cl, same `swap`)."""
    t = t.contiguous()
    F_, sp, Cp = t.shape[0], tuple(t.shape[1:-1]), t.shape[-1]
    P = 1
    for s in sp:
        P *= s
    out = torch.empty((F_, channels) + sp, dtype=torch.float32, device=t.device)
    A, B = swap if swap else (F_, 1)
    L.check(L.lib().dvd_from_channels_last(L.dt(t), L.ptr(t), L.ptr(out), C.c_longlong(F_), channels,
                                           C.c_longlong(P), Cp, A, B, 1 if swap else 0, L.stream()))
    return out


def convert(src, dtype):
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    L.check(L.lib().dvd_convert(L.dt(src), L.ptr(src), L.dt(out), L.ptr(out), C.c_longlong(src.numel()), L.stream()))
    return out


# ------------------------------------------------------------------ weights
class PackedConv:
    """Forward pack wf [ntaps][Cout_tot][Cip] and backward-data pack wd [ntaps][Cip][Cop]."""

    def __init__(self, dtype, cout_tot, cin, ksize, device, need_dgrad=True):
        self.k = _ksize3(ksize)
        self.ntaps = self.k[0] * self.k[1] * self.k[2]
        self.cout, self.cin, self.cip, self.cop = cout_tot, cin, pad8(cin), pad8(cout_tot)
        self.wf = torch.zeros(self.ntaps, cout_tot, self.cip, dtype=dtype, device=device)
        self.wd = torch.zeros(self.ntaps, self.cip, self.cop, dtype=dtype, device=device) if need_dgrad else None

    def fill(self, w, sigma=None, co_off=0):
        """w: fp32 master [Cout_part, Cin, *k]; sigma: device scalar tensor or None."""
        w = w.contiguous()
        L.check(L.lib().dvd_pack_conv_weight(
            L.dt(self.wf), L.ptr(w), L.ptr(sigma), w.shape[0], self.cin, self.ntaps, self.cip, co_off,
            self.cout, self.cop, L.ptr(self.wf), L.ptr(self.wd), self.k[0], self.k[1], self.k[2], L.stream()))
        return self


# ------------------------------------------------------------------ convolution
def _grid(x, ksize, up2):
    """(frames, T, H, W) of the OUTPUT for input x [F,(T,)H,W,C]."""
    if x.dim() == 5:
        F_, T, H, W = x.shape[:4]
    elif x.dim() == 4:
        F_, H, W = x.shape[:3]
        T = 1
    elif x.dim() == 2:
        F_, T, H, W = x.shape[0], 1, 1, 1
    else:
        raise ValueError("expected [F,(T,)H,W,C] or [F,C]")
    if up2:
        H, W = H * 2, W * 2
    return F_, T, H, W


def conv_forward(x, wpack, ksize, cout, *, bias=None, res=None, mask=None, act=L.ACT_NONE, up2=False,
                 relu_in=False, out=None, out_f32=False, nsplit=1, ws=None, slabs=False, cout_pad=None):
    """Direct (nsplit=1) or split-K convolution.  `wpack`: [ntaps][cout][Cp] tensor.  Returns the
    output tensor [F,(T,)H,W,cout_pad] (direct) or the fp32 slabs [nsplit, M, cout] (split-K)."""
    k = _ksize3(ksize)
    F_, T, H, W = _grid(x, ksize, up2)
    Cp = x.shape[-1]
    M = F_ * T * H * W
    d = L.ConvDesc()
    d.dtype, d.frames, d.T, d.H, d.W = L.dt(x), F_, T, H, W
    d.C, d.ldi, d.Cout = Cp, Cp, cout
    d.kt, d.kh, d.kw = k
    d.up2, d.relu_in, d.nsplit, d.act, d.out_f32 = int(up2), int(relu_in), nsplit, act, int(out_f32)
    d.inp, d.w, d.bias = x.data_ptr(), wpack.data_ptr(), (bias.data_ptr() if bias is not None else None)
    nk = k[0] * k[1] * k[2] * ((Cp + (31 if x.dtype == torch.bfloat16 else 15)) // (32 if x.dtype == torch.bfloat16 else 16))
    nsplit = d.nsplit = max(1, min(nsplit, nk))
    if nsplit > 1 or slabs or ws is not None:
        if ws is None:
            ws = torch.empty(nsplit, M, cout, dtype=torch.float32, device=x.device)
        d.ws, d.ldo = ws.data_ptr(), cout
        L.check(L.lib().dvd_conv_forward(C.byref(d), L.stream()))
        return ws
    cp_out = cout_pad or pad8(cout)
    if out is None:
        shape = ((F_,) if x.dim() == 2 else (F_, H, W) if x.dim() == 4 else (F_, T, H, W)) + (cp_out,)
        alloc = torch.zeros if cp_out != cout else torch.empty
        out = alloc(shape, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    d.out, d.ldo = out.data_ptr(), out.shape[-1]
    if res is not None:
        d.res, d.ldres = res.data_ptr(), res.shape[-1]
    if mask is not None:
        d.mask, d.ldmask = mask.data_ptr(), mask.shape[-1]
    L.check(L.lib().dvd_conv_forward(C.byref(d), L.stream()))
    return out


def conv_wgrad(x, dy, dw, ksize, cout, cin_real, *, up2=False, relu_in=False, msplit=0):
    """dw (fp32, reference layout [cout][cin_real][*k], accumulated atomically) += x (*) dy."""
    k = _ksize3(ksize)
    F_, T, H, W = _grid(x, ksize, up2)
    ntaps = k[0] * k[1] * k[2]
    d = L.WgradDesc()
    d.dtype, d.frames, d.T, d.H, d.W = L.dt(x), F_, T, H, W
    d.C, d.ldx, d.Cin_real = x.shape[-1], x.shape[-1], cin_real
    d.Cout, d.Cy, d.ldy = cout, dy.shape[-1], dy.shape[-1]
    d.kt, d.kh, d.kw = k
    d.up2, d.relu_in, d.msplit = int(up2), int(relu_in), msplit
    d.s_co, d.s_ci, d.s_tap = cin_real * ntaps, ntaps, 1
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    L.check(L.lib().dvd_conv_wgrad(C.byref(d), L.stream()))
    return dw
