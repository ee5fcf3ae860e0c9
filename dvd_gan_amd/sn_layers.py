"""Parameter containers that reproduce the reference state_dict layout, plus the spectral-norm and
conditional-batch-norm front ends of the HIP kernels.

Reference: Module/Normalization.py (SpectralNorm :10-64, ConditionalNorm :66-88).  Key layout kept:
  <name>.module.{bias, weight_u, weight_v, weight_bar}        (SpectralNorm wrapper)
  <name>.{bn.running_mean, bn.running_var, bn.num_batches_tracked, embed.weight, embed.bias}
"""
import math
import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import kern as K
from . import lib as L


def _conv_default_init(w, b):
    """nn.Conv*/nn.Linear default: kaiming_uniform(a=sqrt(5)) and bias U(+-1/sqrt(fan_in))."""
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    if b is not None:
        fan_in = w[0].numel()
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(b, -bound, bound)


class ConvParams(nn.Module):
    """Plain (not spectrally normalised) conv parameters: `.weight`, `.bias` -- nn.Conv2d keys."""

    def __init__(self, cin, cout, ksize, init="default"):
        super().__init__()
        self.cin, self.cout, self.ksize = cin, cout, tuple(ksize)
        self.weight = nn.Parameter(torch.empty(cout, cin, *ksize))
        self.bias = nn.Parameter(torch.empty(cout))
        if init == "orthogonal":            # ConvGRU.py:20-26
            nn.init.orthogonal_(self.weight)
            nn.init.zeros_(self.bias)
        elif init == "xavier":              # Discriminators.py:73-76
            nn.init.xavier_uniform_(self.weight)
            nn.init.zeros_(self.bias)
        else:
            _conv_default_init(self.weight, self.bias)


class _SNInner(nn.Module):
    """What the reference leaves inside SpectralNorm.module after _make_params (:43-59)."""

    def __init__(self, shape, bias_n, init_weight=None):
        super().__init__()
        w = torch.empty(*shape)
        b = torch.empty(bias_n) if bias_n else None
        if init_weight is not None:
            init_weight(w)
            if b is not None:
                nn.init.uniform_(b, -1 / math.sqrt(w[0].numel()), 1 / math.sqrt(w[0].numel()))
        else:
            _conv_default_init(w, b)
        if b is not None:
            self.bias = nn.Parameter(b)
        height = shape[0]
        width = w.numel() // height
        u = torch.randn(height)
        v = torch.randn(width)
        self.weight_u = nn.Parameter(u / (u.norm() + 1e-12), requires_grad=False)
        self.weight_v = nn.Parameter(v / (v.norm() + 1e-12), requires_grad=False)
        self.weight_bar = nn.Parameter(w)


class SpectralNormConv(nn.Module):
    """SpectralNorm(nn.Conv2d / nn.Conv3d).  Every call runs one power iteration (also in eval /
    no_grad, quirk 2), then packs W_bar / sigma for the MFMA kernels."""

    def __init__(self, cin, cout, ksize):
        super().__init__()
        self.cin, self.cout, self.ksize = cin, cout, tuple(ksize)
        self.module = _SNInner((cout, cin) + self.ksize, cout)
        self.train_weights = True          # False: treat weights as constants (G step through D)

        self._pre = None                   # (sigma, pack, event) prepared on the SN side stream for the next forward
        self._frag_wants = set()           # fragment-major images ("wf" / "wd") the layer's convolutions asked for so far

    def _alloc(self, dtype, device):
        return (torch.empty(1, dtype=torch.float32, device=device),
                K.PackedConv(dtype, self.cout, self.cin, self.ksize, device, single_fill=True))

    def _sn_and_pack(self, sigma, pack):
        """One power iteration (u, v updated in place), sigma, and W_bar / sigma written into the MFMA operand images."""
        m = self.module
        K.sn_power_iter(m.weight_bar.data, m.weight_u.data, m.weight_v.data, sigma)
        pack.fill(m.weight_bar.data, sigma)

    def forward(self, x, *, res=None, act=L.ACT_NONE, up2=False, relu_in=False, slot=None):
        m = self.module
        if self._pre is not None:          # prepared by prefetch_spectral_norm on the side stream
            sigma, pack, ev = self._pre
            self._pre = None
            torch.cuda.current_stream().wait_event(ev)
        else:
            sigma, pack = self._alloc(x.dtype, x.device)
            self._sn_and_pack(sigma, pack)
        pack.wants = self._frag_wants
        spec = Fn.ConvSpec(self.ksize, self.cout, self.cin, act=act, up2=up2, relu_in=relu_in,
                           sn=(m.weight_u.data, m.weight_v.data))
        spec.sigma = sigma
        spec.pack = pack
        w, b = (m.weight_bar, m.bias) if self.train_weights else (m.weight_bar.detach(), m.bias.detach())
        return Fn.Conv.apply(x, w, b, res, spec, slot)


_SN_STREAMS = {}


def prefetch_spectral_norm(net, dtype):
    """Spectral norm of every SN conv of `net` for the forward that is starting: power iteration + sigma + the
    sigma-normalised MFMA weight images, all issued on a side stream (the reference does this inside each layer's
    forward, Normalization.py:19-31,61-63; u / v advance exactly once per network forward either way, quirk 2).
    Outputs are allocated on the caller's stream; each layer waits for its own event before its convolution.
    Returns the modules so the caller can drop unused preparations (`clear_spectral_norm`)."""
    mods = [m for m in net.modules() if isinstance(m, SpectralNormConv)]
    if not mods or not torch.cuda.is_available() or os.environ.get("DVD_SN_SIDE", "1") == "0":
        return mods
    dev = mods[0].module.weight_bar.device
    main = torch.cuda.current_stream(dev)
    side = _SN_STREAMS.get(dev.index)
    if side is None:
        side = _SN_STREAMS[dev.index] = torch.cuda.Stream(dev)
    bufs = [m._alloc(dtype, dev) for m in mods]          # allocated (and zero-filled) in main-stream order
    if os.environ.get("DVD_SN_BATCHED", "1") != "0":
        # ONE item table, four launches for the whole network (dvd_sn_batched: W^T u, W v, finish, packs) instead of five
        # launches per layer; every layer waits for the same event
        import ctypes as C
        n = len(mods)
        items = (L.SnItem * n)()
        for it, m, (sigma, pack) in zip(items, mods, bufs):
            w = m.module.weight_bar.data
            it.W, it.u, it.v, it.sigma = w.data_ptr(), m.module.weight_u.data_ptr(), m.module.weight_v.data_ptr(), sigma.data_ptr()
            it.wf, it.wd = pack.wf.data_ptr(), (pack.wd.data_ptr() if pack.wd is not None else None)
            it.h, it.w = w.shape[0], w.numel() // w.shape[0]
            it.dtype, it.cout, it.cin, it.ntaps, it.cip, it.cop = L.dt(pack.wf), pack.cout, pack.cin, pack.ntaps, pack.cip, pack.cop
        nfl = C.c_longlong()
        L.check(L.lib().dvd_sn_batched_prepare(items, n, C.byref(nfl)))
        host = torch.empty(C.sizeof(items), dtype=torch.uint8, pin_memory=True)
        C.memmove(host.data_ptr(), C.addressof(items), C.sizeof(items))
        scratch = torch.empty(max(1, nfl.value), dtype=torch.float32, device=dev)
        # the fragment-major images each layer's convolutions used in earlier steps, in one launch behind the packs (they were ~100
        # launches of 6 us per step on the main stream, one in front of the layer's convolution); buffers allocated in main-stream order
        pb = K.PackBatch()
        if dtype == torch.bfloat16 and K.PACK_BATCH:
            for m, (sigma, pack) in zip(mods, bufs):
                for which in sorted(m._frag_wants):
                    if getattr(pack, which) is not None:
                        pb.fragment_major(pack, which)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            table = host.to(dev, non_blocking=True)
            L.check(L.lib().dvd_sn_batched(items, C.c_void_p(table.data_ptr()), n, C.c_void_p(scratch.data_ptr()),
                                           C.c_void_p(side.cuda_stream)))
            pb.run()
            ev = torch.cuda.Event()
            ev.record(side)
        scratch.record_stream(side)
        for m, (sigma, pack) in zip(mods, bufs):
            m._pre = (sigma, pack, ev)
        return mods
    side.wait_stream(main)                               # weights, u, v and the fresh buffers are current
    with torch.cuda.stream(side):
        for m, (sigma, pack) in zip(mods, bufs):
            m._sn_and_pack(sigma, pack)
            ev = torch.cuda.Event()
            ev.record(side)
            m._pre = (sigma, pack, ev)
    return mods


def clear_spectral_norm(mods):
    for m in mods:
        m._pre = None


class PlainConv(nn.Module):
    """nn.Conv2d with reference keys `.weight/.bias` (attention q/k/v)."""

    def __init__(self, cin, cout, ksize, init="default"):
        super().__init__()
        self.p = None
        self.cin, self.cout, self.ksize = cin, cout, tuple(ksize)
        self.weight = nn.Parameter(torch.empty(cout, cin, *ksize))
        self.bias = nn.Parameter(torch.empty(cout))
        if init == "xavier":
            nn.init.xavier_uniform_(self.weight)
            nn.init.zeros_(self.bias)
        else:
            _conv_default_init(self.weight, self.bias)


class ConditionalNorm(nn.Module):
    """Normalization.py:66-88.  `embed` is an nn.Linear(n_condition, 2*C) with the reference's
    (axis-confused) init: weight[:, :C] ~ N(1, .02), weight[:, C:] = 0 (quirk, :75-76)."""

    def __init__(self, in_channel, n_condition):
        super().__init__()
        self.in_channel = in_channel
        self.bn = nn.Module()
        self.bn.register_buffer("running_mean", torch.zeros(in_channel))
        self.bn.register_buffer("running_var", torch.ones(in_channel))
        self.bn.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.embed = nn.Module()
        w = torch.empty(in_channel * 2, n_condition)
        b = torch.empty(in_channel * 2)
        _conv_default_init(w, b)
        w[:, :in_channel].normal_(1, 0.02)
        w[:, in_channel:].zero_()
        self.embed.weight = nn.Parameter(w)
        self.embed.bias = nn.Parameter(b)
        self.replicas = None               # (world, all_reduce_sum_) when batch statistics span all replicas
        self._sums = None                  # persistent fp64 workspace of the statistics kernel (left zeroed by dvd_bn_finalize)
        self.count_batches = True          # False: a caller (the Generator) advances num_batches_tracked for all layers at once

    def forward(self, x, cond, samp, relu=True, tok=None, slot=None):
        """x: channels-last [frames, H, W, Cp]; cond: fp32 [B, n_condition]; samp: int32 [frames].  tok / slot: functional.GradSlot."""
        gb = Fn.LinearF32.apply(cond, self.embed.weight, self.embed.bias)
        if self.training and self.count_batches:
            self.bn.num_batches_tracked += 1
        if self._sums is None or self._sums.device != x.device:
            self._sums = torch.zeros(L.BN_NREP * 2 * self.in_channel, dtype=torch.float64, device=x.device)
        return Fn.CondBatchNorm.apply(x, gb, samp, self.in_channel, relu, self.training, self.bn.running_mean,
                                      self.bn.running_var, 1e-5, 0.1, self.replicas, self._sums, tok, slot)
