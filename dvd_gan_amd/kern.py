"""Tensor-level wrappers over the C ABI (one Python function per entry point).

Internal activation layout: channels-last, [frames, (T,) H, W, Cp] with Cp = channels padded to a
multiple of 8, storage dtype torch.float32 (exact mode) or torch.bfloat16.
"""
import ctypes as C
import os

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

    def __init__(self, dtype, cout_tot, cin, ksize, device, need_dgrad=True, single_fill=False, covered=False):
        """single_fill: ONE fill() call will cover all cout_tot output channels -- it writes every element of both
        images (zeros in the padded input channels) except the padded output columns of `wd`, so only a `wd` with
        cout_tot % 8 != 0 needs the zero fill.  covered: the caller's SEVERAL fill() calls together cover every output
        channel (the fused gate packs of a ConvGRU layer): same consequence."""
        self.k = _ksize3(ksize)
        self.ntaps = self.k[0] * self.k[1] * self.k[2]
        self.cout, self.cin, self.cip, self.cop = cout_tot, cin, pad8(cin), pad8(cout_tot)
        af = torch.empty if (single_fill or covered) else torch.zeros
        ad = torch.empty if ((single_fill or covered) and self.cop == cout_tot) else torch.zeros
        self.wf = af(self.ntaps, cout_tot, self.cip, dtype=dtype, device=device)
        self.wd = ad(self.ntaps, self.cip, self.cop, dtype=dtype, device=device) if need_dgrad else None

    def fragment_major(self, which="wf"):
        """The forward ("wf") or backward-data ("wd") image once more in fragment-major order (dvd_conv_fragment_major), for
        the convolution kernel that reads its weight operand straight from L2; built on first use, after the last fill()."""
        wants = getattr(self, "wants", None)
        if wants is not None:
            wants.add(which)              # (the owner prepares this image ahead of time from the next forward on: sn_layers.prefetch_spectral_norm)
        q = getattr(self, which + "q", None)
        if q is None:
            src = getattr(self, which)
            cout, c = (self.cout, self.cip) if which == "wf" else (self.cip, self.cop)
            n = L.lib().dvd_conv_fragment_major_bytes(self.ntaps, cout, c)
            q = torch.empty(n // 2, dtype=src.dtype, device=src.device)
            L.check(L.lib().dvd_conv_fragment_major(L.dt(src), L.ptr(src), L.ptr(q), self.ntaps, cout, c, L.stream()))
            setattr(self, which + "q", q)
        return q

    def fill(self, w, sigma=None, co_off=0, ci_off=0):
        """w: fp32 master [Cout_part, Cin_tot, *k] of which input channels [ci_off, ci_off+cin) are
        packed as output rows [co_off, co_off+Cout_part); sigma: device scalar tensor or None."""
        assert w.is_contiguous() and w.dtype == torch.float32
        L.check(L.lib().dvd_pack_conv_weight(
            L.dt(self.wf), L.ptr(w), L.ptr(sigma), w.shape[0], self.cin, self.ntaps, self.cip, co_off,
            self.cout, self.cop, L.ptr(self.wf), L.ptr(self.wd), self.k[0], self.k[1], self.k[2],
            ci_off, w.shape[1], L.stream()))
        return self


PACK_BATCH = os.environ.get("DVD_PACK_BATCH", "1") != "0"
# timing-only experiment (WRONG results): 1 = BN statistics from 4096 rows, 2 = CBN backward sums from two frames -- what folding
# those two passes into the neighbouring convolutions' epilogues could return at most (profiles/HISTORY.md, round 6)
EXP_CBN = int(os.environ.get("DVD_EXP_CBN", "0"))
EXP_SKIP = int(os.environ.get("DVD_EXP_SKIP", "0"))      # 1 = skip the 1 x 1 convolutions on >= 128 k rows (garbage results)


class PackBatch:
    """Deferred PackedConv.fill / fragment_major calls issued as ONE launch each (dvd_pack_conv_weight_batched,
    dvd_conv_fragment_major_batched): the 18 gate packs and up to 18 fragment-major images of a three-layer ConvGRU were 36+
    launches of 6-18 us on the stream in front of its first convolution."""

    def __init__(self):
        self.fills, self.frags, self.dtype = [], [], None

    def fill(self, pk, w, sigma=None, co_off=0, ci_off=0):
        assert w.is_contiguous() and w.dtype == torch.float32
        it = L.PackItem()
        it.w, it.sigma, it.wf, it.wd = w.data_ptr(), L.ptr(sigma), L.ptr(pk.wf), L.ptr(pk.wd)
        it.Cout, it.Cin, it.ntaps, it.Cip, it.co_off = w.shape[0], pk.cin, pk.ntaps, pk.cip, co_off
        it.co_tot_f, it.co_tot_d = pk.cout, pk.cop
        it.kt, it.kh, it.kw = pk.k
        it.ci_off, it.ci_tot = ci_off, w.shape[1]
        assert self.dtype in (None, pk.wf.dtype)
        self.dtype = pk.wf.dtype
        self.fills.append((it, w, sigma, pk))
        return self

    def fragment_major(self, pk, which="wf"):
        """Schedules the fragment-major image of `pk.<which>` (built by run(), after the fills) and returns the tensor it will be in."""
        q = getattr(pk, which + "q", None)
        if q is None:
            src = getattr(pk, which)
            cout, c = (pk.cout, pk.cip) if which == "wf" else (pk.cip, pk.cop)
            n = L.lib().dvd_conv_fragment_major_bytes(pk.ntaps, cout, c)
            q = torch.empty(n // 2, dtype=src.dtype, device=src.device)
            it = L.FragItem()
            it.w, it.wq, it.ntaps, it.Cout, it.C = src.data_ptr(), q.data_ptr(), pk.ntaps, cout, c
            self.frags.append((it, src, q))
            setattr(pk, which + "q", q)
        return q

    def run(self):
        dt_ = L.BF16 if self.dtype == torch.bfloat16 else L.F32
        if not PACK_BATCH:                # A/B aid: one launch per item, as before round 6
            for it, *_ in self.fills:
                vp = C.c_void_p
                L.check(L.lib().dvd_pack_conv_weight(dt_, vp(it.w), vp(it.sigma), it.Cout, it.Cin, it.ntaps, it.Cip, it.co_off, it.co_tot_f,
                                                     it.co_tot_d, vp(it.wf), vp(it.wd), it.kt, it.kh, it.kw, it.ci_off, it.ci_tot, L.stream()))
            for it, *_ in self.frags:
                L.check(L.lib().dvd_conv_fragment_major(L.BF16, C.c_void_p(it.w), C.c_void_p(it.wq), it.ntaps, it.Cout, it.C, L.stream()))
        else:
            if self.fills:
                arr = (L.PackItem * len(self.fills))(*[f[0] for f in self.fills])
                L.check(L.lib().dvd_pack_conv_weight_batched(dt_, arr, len(self.fills), L.stream()))
            if self.frags:
                arr = (L.FragItem * len(self.frags))(*[f[0] for f in self.frags])
                L.check(L.lib().dvd_conv_fragment_major_batched(L.BF16, arr, len(self.frags), L.stream()))
        self.fills, self.frags = [], []


# ------------------------------------------------------------------ convolution
def wants_fragment_major(dtype, frames, H, W, cin, cout, k, nsplit=1):
    """True when a [frames, H, W, C] -> cout convolution with k x k taps (nsplit K slices) runs through the kernel that takes a
    fragment-major weight image (dvd_conv_wants_fragment_major)."""
    if dtype != torch.bfloat16:
        return False
    d = L.ConvDesc()
    d.dtype, d.frames, d.T, d.H, d.W = L.BF16, frames, 1, H, W
    d.C, d.ldi, d.Cout, d.ldo = cin, cin, cout, cout
    d.kt, d.kh, d.kw, d.nsplit = 1, k, k, max(1, nsplit)
    d.inp = d.out = d.w = 1               # (never dereferenced: geometry query)
    if nsplit > 1:
        d.ws = 1
    return bool(L.lib().dvd_conv_wants_fragment_major(C.byref(d)))


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


def _thin_image(wpack, kt):
    """The [tap row][k half][channel block][lane][8] image of a [kt*9][64][8] pack for the thin-input convolution kernel
    (dvd_conv_thin_image); cached on the pack tensor (packs are rebuilt whenever the weights change)."""
    img = getattr(wpack, "_thin_img", None)
    if img is None:
        n = L.lib().dvd_conv_thin_image_bytes(kt)
        img = torch.empty(n // 2, dtype=wpack.dtype, device=wpack.device)
        L.check(L.lib().dvd_conv_thin_image(L.ptr(wpack), L.ptr(img), kt, L.stream()))
        wpack._thin_img = img
    return img


def _thin_out_image(wpack):
    """The [tap][16-channel step][lane][8] image of a [9][rows <= 8][64] pack for the thin-output convolution kernel."""
    img = getattr(wpack, "_thin_img", None)
    if img is None:
        img = torch.empty(L.lib().dvd_conv_thin_out_image_bytes() // 2, dtype=wpack.dtype, device=wpack.device)
        L.check(L.lib().dvd_conv_thin_out_image(L.ptr(wpack), L.ptr(img), wpack.shape[1], L.stream()))
        wpack._thin_img = img
    return img


def conv_forward(x, wpack, ksize, cout, *, bias=None, res=None, mask=None, act=L.ACT_NONE, up2=False,
                 relu_in=False, out=None, out_f32=False, nsplit=1, ws=None, slabs=False, cout_pad=None, res_up2=False, wq=None,
                 pool2=False):
    """Direct (nsplit=1) or split-K convolution.  `wpack`: [ntaps][cout][Cp] tensor.  Returns the
    output tensor [F,(T,)H,W,cout_pad] (direct) or the fp32 slabs [nsplit, M, cout] (split-K).
    pool2: return the 2 x 2 SUMS of the result on the half-size grid [F, H/2, W/2, cout_pad] (`mask` lives there too), formed in the
    kernel's epilogue -- or None when this request is not served that way (the caller then pools a full-size result)."""
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
    if res is not None:
        d.res, d.ldres, d.res_up2 = res.data_ptr(), res.shape[-1], int(res_up2)
    if mask is not None:
        d.mask, d.ldmask = mask.data_ptr(), mask.shape[-1]
    if callable(wq):                      # lazily built fragment-major image: only when this request runs through that kernel
        d.nsplit, d.out = max(1, nsplit), 1          # (placeholders for the geometry query; the real pointers are set below)
        d.ldo = out.shape[-1] if out is not None else (cout_pad or pad8(cout))
        d.ws = 1 if (nsplit > 1 or slabs or ws is not None) else None
        want = L.lib().dvd_conv_wants_fragment_major(C.byref(d)) if x.dtype == torch.bfloat16 else 0
        wq = wq() if want == 1 else _thin_image(wpack, k[0]) if want == 2 else _thin_out_image(wpack) if want == 3 else None
        d.wq_kind = want if wq is not None else 0
    d.wq = wq.data_ptr() if wq is not None else None
    d.out = d.ws = None
    if pool2:
        d.pool2 = 1
        d.ldo = cout_pad or pad8(cout)
        if x.dim() != 4 or nsplit > 1 or slabs or ws is not None or not L.lib().dvd_conv_pool2_ok(C.byref(d)):
            return None
        shape = (F_, H // 2, W // 2, d.ldo)
        alloc = torch.zeros if d.ldo != cout else torch.empty
        out = alloc(shape, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
        d.out = out.data_ptr()
        L.check(L.lib().dvd_conv_forward(C.byref(d), L.stream()))
        return out
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
    if EXP_SKIP & 1 and k == (1, 1, 1) and M >= (1 << 17):       # timing experiment: the large 1 x 1 convolutions are not launched
        return out
    L.check(L.lib().dvd_conv_forward(C.byref(d), L.stream()))
    return out


def conv_wgrad(x, dy, dw, ksize, cout, cin_real, *, up2=False, relu_in=False, msplit=0, dy_col=0,
               dw_ci_off=0, dw_ci_tot=None, frames=None, x_row0=0, dy_row0=0, dbias=None, overwrite=False):
    """dw[co][dw_ci_off + ci][*k] += sum_rows dy[row][dy_col + co] * x[shifted row][ci]   (fp32 atomics)
    dw: fp32 master-layout tensor [cout][dw_ci_tot][*k].  x / dy may be row-sliced views given by
    (tensor, first frame): `frames` frames starting at frame x_row0 of x and dy_row0 of dy."""
    k = _ksize3(ksize)
    F_, T, H, W = _grid(x, ksize, up2)
    if frames is not None:
        F_ = frames
    ntaps = k[0] * k[1] * k[2]
    esz = x.element_size()
    ci_tot = dw_ci_tot if dw_ci_tot is not None else cin_real
    d = L.WgradDesc()
    d.dtype, d.frames, d.T, d.H, d.W = L.dt(x), F_, T, H, W
    d.C, d.ldx, d.Cin_real = x.shape[-1], x.shape[-1], cin_real
    cy = min(pad8(cout), dy.shape[-1] - dy_col)
    d.Cout, d.Cy, d.ldy = cout, cy, dy.shape[-1]
    d.kt, d.kh, d.kw = k
    d.up2, d.relu_in, d.msplit = int(up2), int(relu_in), msplit
    d.s_co, d.s_ci, d.s_tap = ci_tot * ntaps, ntaps, 1
    rows_in = T * (H // 2 if up2 else H) * (W // 2 if up2 else W)
    d.x = x.data_ptr() + x_row0 * rows_in * x.shape[-1] * esz
    d.dy = dy.data_ptr() + (dy_row0 * T * H * W * dy.shape[-1] + dy_col) * esz
    d.dw = dw.data_ptr() + dw_ci_off * ntaps * 4
    d.dbias = dbias.data_ptr() if dbias is not None else None      # fp32 [cout], accumulated
    d.overwrite = int(overwrite)                                    # dw = result (dense dw only): `dw` may be torch.empty
    nws = L.lib().dvd_conv_wgrad_ws_floats(C.byref(d))              # >0: deterministic two-phase reduction
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws > 0 else None
    d.ws = ws.data_ptr() if ws is not None else None
    L.check(L.lib().dvd_conv_wgrad(C.byref(d), L.stream()))
    return dw


# ------------------------------------------------------------------ pointwise wrappers
def _ll(v):
    return C.c_longlong(int(v))


def _f(v):
    return C.c_float(float(v))


def bn_stats(x, C_real, training, eps, momentum, run_mean, run_var, replicas=None, sums=None):
    """-> (mean, rstd) fp32 [C_real]; updates the running buffers in training mode.
    sums: optional PERSISTENT zeroed fp64 workspace [BN_NREP * 2 * C_real] of the caller -- dvd_bn_finalize leaves it zeroed
    again, so no fill is launched per call.
    replicas: None, or (world, all_reduce_sum_) for cross-replica statistics -- the [BN_NREP][2C] fp64 sums are added over the
    replicas before mean / variance are formed from world * rows samples."""
    ld = x.shape[-1]
    rows = x.numel() // ld
    mean = torch.empty(C_real, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    keep = sums is not None and training
    if not keep:
        sums = None
    try:
        if training:
            if sums is None:
                sums = torch.zeros(L.BN_NREP * 2 * C_real, dtype=torch.float64, device=x.device)
            if EXP_CBN & 1:               # timing experiment (wrong statistics): the pass over x reads 4096 rows only
                rows = min(rows, 4096)
            L.check(L.lib().dvd_bn_stats(L.dt(x), L.ptr(x), _ll(rows), C_real, ld, L.ptr(sums), L.stream()))
            if replicas is not None:
                replicas[1](sums)
                rows *= replicas[0]
        L.check(L.lib().dvd_bn_finalize(L.ptr(sums), _ll(rows), C_real, _f(eps), _f(momentum), int(training),
                                        L.ptr(mean), L.ptr(rstd), L.ptr(run_mean), L.ptr(run_var), int(keep), L.stream()))
    except Exception:
        if keep:
            sums.zero_()                  # a failed call must not leave a dirty workspace behind
        raise
    return mean, rstd


def cbn_apply(x, C_real, mean, rstd, gb, samp, relu):
    y = torch.empty_like(x)
    frames = x.shape[0]
    P = x.numel() // (frames * x.shape[-1])
    L.check(L.lib().dvd_cbn_apply(L.dt(x), L.ptr(x), L.ptr(y), _ll(frames), P, C_real, x.shape[-1], L.ptr(mean),
                                  L.ptr(rstd), L.ptr(gb), L.ptr(samp), int(relu), L.stream()))
    return y


def cbn_backward(g, a, x, C_real, mean, rstd, gb, samp, relu, replicas=None):
    """-> (dx, dgb[B][2C]).  replicas: as in bn_stats -- the two per-channel sums of the batch-norm backward are added
    over the replicas between the reduce and the apply stage."""
    dx = torch.empty_like(x)
    frames = x.shape[0]
    P = x.numel() // (frames * x.shape[-1])
    B = gb.shape[0]
    dgb = torch.zeros_like(gb)
    s12 = torch.empty(2 * C_real, dtype=torch.float32, device=x.device)
    part = torch.empty(L.lib().dvd_cbn_backward_ws_floats(_ll(frames), P, C_real), dtype=torch.float32, device=x.device)   # fixed-order dgb
    if EXP_CBN & 2 and replicas is None:      # timing experiment (wrong sums): the reduce pass reads two frames only
        L.check(L.lib().dvd_cbn_backward_reduce(L.dt(x), L.ptr(g), L.ptr(a), L.ptr(x), _ll(min(frames, 2)), P, C_real, x.shape[-1],
                                                L.ptr(mean), L.ptr(rstd), L.ptr(gb), L.ptr(samp), B, L.ptr(dgb), L.ptr(s12),
                                                int(relu), L.ptr(part), L.stream()))
        L.check(L.lib().dvd_cbn_backward_apply(L.dt(x), L.ptr(g), L.ptr(a), L.ptr(x), L.ptr(dx), _ll(frames), P, C_real,
                                               x.shape[-1], L.ptr(mean), L.ptr(rstd), L.ptr(gb), L.ptr(samp), L.ptr(s12),
                                               _ll(frames * P), int(relu), L.stream()))
        return dx, dgb
    if replicas is None:
        L.check(L.lib().dvd_cbn_backward(L.dt(x), L.ptr(g), L.ptr(a), L.ptr(x), L.ptr(dx), _ll(frames), P, C_real,
                                         x.shape[-1], L.ptr(mean), L.ptr(rstd), L.ptr(gb), L.ptr(samp), B, L.ptr(dgb),
                                         L.ptr(s12), int(relu), L.ptr(part), L.stream()))
        return dx, dgb
    L.check(L.lib().dvd_cbn_backward_reduce(L.dt(x), L.ptr(g), L.ptr(a), L.ptr(x), _ll(frames), P, C_real, x.shape[-1],
                                            L.ptr(mean), L.ptr(rstd), L.ptr(gb), L.ptr(samp), B, L.ptr(dgb), L.ptr(s12),
                                            int(relu), L.ptr(part), L.stream()))
    replicas[1](s12)
    L.check(L.lib().dvd_cbn_backward_apply(L.dt(x), L.ptr(g), L.ptr(a), L.ptr(x), L.ptr(dx), _ll(frames), P, C_real,
                                           x.shape[-1], L.ptr(mean), L.ptr(rstd), L.ptr(gb), L.ptr(samp), L.ptr(s12),
                                           _ll(frames * P * replicas[0]), int(relu), L.stream()))
    return dx, dgb


def _grid4(x):
    if x.dim() == 5:
        return x.shape[0], x.shape[1], x.shape[2], x.shape[3]
    return x.shape[0], 1, x.shape[1], x.shape[2]


def pool(x, pt=1, scale=None, mask=None):
    """(pt,2,2) window sum * scale (default: average); mask (shape of the result): zero where mask <= 0."""
    F_, T, H, W = _grid4(x)
    To, Ho, Wo = T // pt, H // 2, W // 2
    scale = (1.0 / (4 * pt)) if scale is None else scale
    shape = (F_, To, Ho, Wo, x.shape[-1]) if x.dim() == 5 else (F_, Ho, Wo, x.shape[-1])
    y = torch.empty(shape, dtype=x.dtype, device=x.device)
    if mask is not None:
        assert mask.shape == y.shape and mask.dtype == y.dtype and not ((H | W) & 1)
        L.check(L.lib().dvd_pool_masked(L.dt(x), L.ptr(x), L.ptr(mask), L.ptr(y), _ll(F_), To, Ho, Wo, x.shape[-1], pt, _f(scale),
                                        L.stream()))
        return y
    L.check(L.lib().dvd_pool(L.dt(x), L.ptr(x), L.ptr(y), _ll(F_), To, Ho, Wo, H, W, x.shape[-1], pt, _f(scale), L.stream()))
    return y


def unpool(x, pt=1, scale=None, out_hw=None):
    """nearest replication to (T*pt, 2H, 2W) times scale (default 1/(4*pt): gradient of avg-pool).  out_hw = (2H+1 | 2H, 2W+1 |
    2W): the grid a floored pooling started from -- its odd last line / column gets zeros."""
    F_, T, H, W = _grid4(x)
    To, Ho, Wo = T * pt, H * 2, W * 2
    if out_hw is not None:
        Ho, Wo = out_hw
        assert Ho // 2 == H and Wo // 2 == W
    scale = (1.0 / (4 * pt)) if scale is None else scale
    shape = (F_, To, Ho, Wo, x.shape[-1]) if x.dim() == 5 else (F_, Ho, Wo, x.shape[-1])
    y = torch.empty(shape, dtype=x.dtype, device=x.device)
    L.check(L.lib().dvd_unpool(L.dt(x), L.ptr(x), L.ptr(y), _ll(F_), To, Ho, Wo, x.shape[-1], pt, _f(scale), L.stream()))
    return y


def colsum(x, C_real, out=None):
    ld = x.shape[-1]
    if out is None:
        out = torch.zeros(C_real, dtype=torch.float32, device=x.device)
    L.check(L.lib().dvd_colsum(L.dt(x), L.ptr(x), _ll(x.numel() // ld), C_real, ld, L.ptr(out), L.stream()))
    return out


def add(a, b):
    out = torch.empty_like(a)
    L.check(L.lib().dvd_add(L.dt(a), L.ptr(a), L.ptr(b), L.ptr(out), _ll(a.numel()), L.stream()))
    return out


def sum_leading(x):
    out = torch.empty(x.shape[1:], dtype=x.dtype, device=x.device)
    L.check(L.lib().dvd_sum_leading(L.dt(x), L.ptr(x), L.ptr(out), x.shape[0], _ll(out.numel()), L.stream()))
    return out


def act_backward(dy, y, act):
    dx = torch.empty_like(dy)
    L.check(L.lib().dvd_act_backward(L.dt(dy), L.ptr(dy), L.ptr(y), L.ptr(dx), _ll(dy.numel()), act, L.stream()))
    return dx


def vid_downsample_raw(src, backward, shape_fwd_in):
    B, T, Cc, H, W = shape_fwd_in
    out = torch.empty((B, T, Cc, H, W) if backward else (B, Cc, T, H // 2, W // 2), dtype=torch.float32,
                      device=src.device)
    L.check(L.lib().dvd_vid_downsample(L.ptr(src), L.ptr(out), B, T, Cc, H, W, int(backward), L.stream()))
    return out


def row_copy(src, idx, nrows_out, L_, scatter, out=None):
    """gather: out[r] = src[idx[r]] ; scatter: out[idx[r]] = src[r] (out pre-zeroed, nrows_out rows)."""
    if out is None:
        out = (torch.zeros if scatter else torch.empty)(nrows_out, L_, dtype=torch.float32, device=src.device)
    L.check(L.lib().dvd_row_copy(L.ptr(src), L.ptr(out), L.ptr(idx), _ll(idx.numel()), _ll(L_), int(scatter), L.stream()))
    return out


# ------------------------------------------------------------------ spectral norm / small fp32 layers
def sn_power_iter(w_bar, u, v, sigma=None):
    """In-place u, v update; returns the 1-element sigma tensor (a fresh one unless `sigma` is given)."""
    if sigma is None:
        sigma = torch.empty(1, dtype=torch.float32, device=w_bar.device)
    h = w_bar.shape[0]
    L.check(L.lib().dvd_sn_power_iter(L.ptr(w_bar), h, w_bar.numel() // h, L.ptr(u), L.ptr(v), L.ptr(sigma), L.stream()))
    return sigma


def sn_backward(G, w_bar, u, v, sigma, out=None):
    """dL/dW_bar from G = dL/d(W_bar / sigma); `out`: fp32 buffer the result is ADDED to (default: a fresh zero tensor)."""
    dW = torch.zeros_like(w_bar) if out is None else out
    scratch = torch.empty(L.SN_SCRATCH, dtype=torch.float32, device=w_bar.device)
    h = w_bar.shape[0]
    L.check(L.lib().dvd_sn_backward(L.ptr(G), L.ptr(w_bar), L.ptr(u), L.ptr(v), L.ptr(sigma), h, w_bar.numel() // h,
                                    L.ptr(dW), L.ptr(scratch), L.stream()))
    return dW


def linear_forward(inp, W, bias):
    B, K = inp.shape
    J = W.shape[0]
    out = torch.empty(B, J, dtype=torch.float32, device=inp.device)
    L.check(L.lib().dvd_linear_forward(L.ptr(inp), L.ptr(W), L.ptr(bias), L.ptr(out), B, K, J, L.stream()))
    return out


def linear_backward(dout, inp, W, need_in, need_w, has_bias, dW=None, db=None):
    """dW / db given: fp32 buffers the kernels ADD into (persistent .grad); else fresh zero tensors."""
    B, K = inp.shape
    J = W.shape[0]
    din = torch.empty_like(inp) if need_in else None
    if dW is None:
        dW = torch.zeros_like(W) if need_w else None
        db = torch.zeros(J, dtype=torch.float32, device=inp.device) if (need_w and has_bias) else None
    L.check(L.lib().dvd_linear_backward(L.ptr(dout), L.ptr(inp), L.ptr(W), L.ptr(din), 0, L.ptr(dW), L.ptr(db), B, K, J,
                                        L.stream()))
    return din, dW, db


def embedding_backward(dout, idx, nrows, dW=None):
    if dW is None:
        dW = torch.zeros(nrows, dout.shape[1], dtype=torch.float32, device=dout.device)
    L.check(L.lib().dvd_embedding_backward(L.ptr(dout), L.ptr(idx), L.ptr(dW), _ll(dout.shape[0]), dout.shape[1], L.stream()))
    return dW


def adam_step(p, g, m, v, lr, b1, b2, eps, step):
    L.check(L.lib().dvd_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), _ll(p.numel()), _f(lr), _f(b1), _f(b2),
                                  _f(eps), int(step), L.stream()))
