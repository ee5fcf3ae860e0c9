"""Generator of DVD-GAN on the HIP kernels -- same constructor / forward signature, same
state_dict keys and the same results (incl. quirks) as Module/Generator.py:13-120,
Module/ConvGRU.py and Module/GResBlock.py, but organised for the hardware:

  * frames are kept t-major ([T][B][H][W][C], channels-last) through the whole network, so every
    time step of a ConvGRU is one contiguous GEMM operand and no permute/stack/cat is ever run;
  * each ConvGRU is evaluated layer by layer: the x-half of all gate convolutions is one batched
    convolution over all T steps, only the h-half stays on the serial chain (csrc/gru.hip);
  * the reference's condition mis-ordering (Generator.py:103-110) becomes an index table `samp`.
"""
import torch
import torch.nn as nn

from . import dist as D
from . import functional as Fn
from . import kern as K
from . import lib as L
from .sn_layers import ConditionalNorm, ConvParams, SpectralNormConv, clear_spectral_norm, prefetch_spectral_norm


import os as _os
_SC_LOWRES = _os.environ.get("DVD_SC_LOWRES", "1") != "0"      # A/B aid: 0 = shortcut conv on the upsampled grid (the reference's order)


class ConvGRUCell(nn.Module):
    """Parameter holder with the reference keys {reset,update,out}_gate.{weight,bias} (ConvGRU.py:16-26)."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        self.input_size, self.hidden_size, self.kernel_size = input_size, hidden_size, kernel_size
        k = (kernel_size, kernel_size)
        self.reset_gate = ConvParams(input_size + hidden_size, hidden_size, k, "orthogonal")
        self.update_gate = ConvParams(input_size + hidden_size, hidden_size, k, "orthogonal")
        self.out_gate = ConvParams(input_size + hidden_size, hidden_size, k, "orthogonal")

    def run(self, x, T, shared_x, h0=None):
        return Fn.ConvGRULayer.apply(x, self.update_gate.weight, self.update_gate.bias, self.reset_gate.weight,
                                     self.reset_gate.bias, self.out_gate.weight, self.out_gate.bias, T, shared_x, h0,
                                     not torch.is_grad_enabled())


class ConvGRU(nn.Module):
    """ConvGRU.py:57-133.  `run` processes ALL T steps: layer l over the whole sequence, then layer
    l+1 (identical data flow: layer l at step t needs layer l-1 at step t and itself at t-1)."""

    def __init__(self, input_size, hidden_sizes, kernel_sizes, n_layers):
        super().__init__()
        if not isinstance(hidden_sizes, list):
            hidden_sizes = [hidden_sizes] * n_layers
        if not isinstance(kernel_sizes, list):
            kernel_sizes = [kernel_sizes] * n_layers
        assert len(hidden_sizes) == n_layers, "`hidden_sizes` must have the same length as n_layers"
        assert len(kernel_sizes) == n_layers, "`kernel_sizes` must have the same length as n_layers"
        self.input_size, self.hidden_sizes, self.kernel_sizes, self.n_layers = input_size, hidden_sizes, kernel_sizes, n_layers
        self.cells = nn.ModuleList(
            ConvGRUCell(input_size if i == 0 else hidden_sizes[i - 1], hidden_sizes[i], kernel_sizes[i])
            for i in range(n_layers))

    def run(self, x, T, shared_x, hidden=None):
        """x: [T*B,S,S,C] t-major (or [B,S,S,C] if shared_x).  Returns the list of per-layer
        sequences [T*B,S,S,h_l] (the reference returns the last step's list per call)."""
        if Fn.ConvGRUStack.usable(x, self.cells):      # layer wavefront: all layers in one pass of grouped launches
            flat = []
            for c in self.cells:
                flat += [c.update_gate.weight, c.update_gate.bias, c.reset_gate.weight, c.reset_gate.bias, c.out_gate.weight,
                         c.out_gate.bias]
            flat += [None if hidden is None else hidden[i] for i in range(self.n_layers)]
            return list(Fn.ConvGRUStack.apply(x, T, shared_x, not torch.is_grad_enabled(), self.n_layers, *flat))
        outs = []
        for i, cell in enumerate(self.cells):
            x = cell.run(x, T, shared_x and i == 0, None if hidden is None else hidden[i])
            outs.append(x)
        return outs


class GResBlock(nn.Module):
    """GResBlock.py:9-86 with bn=True, downsample_factor=1 (the only configuration the generator uses)."""

    def __init__(self, in_channel, out_channel, n_class=96, upsample_factor=2):
        super().__init__()
        self.upsample_factor = upsample_factor
        self.conv0 = SpectralNormConv(in_channel, out_channel, (3, 3))
        self.conv1 = SpectralNormConv(out_channel, out_channel, (3, 3))
        self.conv_sc = SpectralNormConv(in_channel, out_channel, (1, 1))
        self.CBNorm1 = ConditionalNorm(in_channel, n_class)
        self.CBNorm2 = ConditionalNorm(out_channel, n_class)

    def run(self, x, cond, samp):
        up = self.upsample_factor != 1
        # shortcut (GResBlock.py:71-73: upsample, then the 1x1 conv): a 1x1 conv commutes with a nearest upsample value for
        # value, so it runs on the SMALL grid (a quarter of the rows) and conv1's epilogue reads it through the upsample.
        # x feeds both branches: the shortcut's backward-data conv adds the main branch's input gradient in its epilogue
        # (functional.GradSlot) instead of leaving the sum to an elementwise pass of the autograd engine.
        lowres = not up or _SC_LOWRES
        slot = Fn.GradSlot() if (lowres and torch.is_grad_enabled() and x.requires_grad) else None
        skip = self.conv_sc(x, up2=up and not _SC_LOWRES, slot=slot)
        tok = None
        if slot is not None:
            skip, tok = skip
        a1 = self.CBNorm1(x, cond, samp, relu=True, tok=tok, slot=slot)
        c0 = self.conv0(a1, up2=up)
        a2 = self.CBNorm2(c0, cond, samp, relu=True)
        return self.conv1(a2, res=skip)


class Generator(nn.Module):
    """Generator(in_dim=120, latent_dim=4, n_class=4, ch=32, n_frames=48, hierar_flag=False)
    .forward(z [B,in_dim] f32, class_id [B] i64) -> [B, T, 3, 16*latent_dim, 16*latent_dim] f32."""

    def __init__(self, in_dim=120, latent_dim=4, n_class=4, ch=32, n_frames=48, hierar_flag=False,
                 compute_dtype=torch.bfloat16, self_attn=False, sep_attn=False):
        """self_attn / sep_attn switch on the two attention blocks the reference defines and imports (Generator.py:10) but
        leaves commented out: `self.self_attn = SelfAttention(8 * ch)` (:29) over the (T, ld, ld) latent clip after the first
        ConvGRU, and `SeparableAttn(4 * ch)` (:34, after the block that produces 4*ch channels) over the (T, 8 ld, 8 ld) clip
        after module 8.  Parameter keys: `self_attn.*`, `sep_attn.model.{0,1,2}.*` (absent when off: reference checkpoints load)."""
        super().__init__()
        if hierar_flag:
            raise NotImplementedError("hierar_flag=True is broken in the reference (Generator.py:66,109)")
        if ch < 2 or ch % 2:
            raise ValueError(f"ch={ch}: the ConvGRU kernels keep hidden states in 16-byte channel vectors, so the smallest "
                             "hidden size 4*ch must be a multiple of 8 (ch even)")
        if latent_dim < 1:
            raise ValueError(f"latent_dim={latent_dim}")
        # (power-of-two latent_dim -- 4 -> 64x64, 8 -> 128x128 clips -- runs the LDS-staged kernels; any other value, e.g. 3 or 6
        #  -> 48x48 / 96x96, the tap-by-tap kernels with division indexing: same results, lower throughput)
        self.in_dim, self.latent_dim, self.n_class, self.ch, self.n_frames = in_dim, latent_dim, n_class, ch, n_frames
        self.hierar_flag = hierar_flag
        self.compute_dtype = compute_dtype
        self.embedding = nn.Embedding(n_class, in_dim)
        self.affine_transfrom = nn.Linear(in_dim * 2, latent_dim * latent_dim * 8 * ch)
        c8, c4, c2 = 8 * ch, 4 * ch, 2 * ch
        nc = in_dim * 2
        self.conv = nn.ModuleList([
            ConvGRU(c8, [c8, 2 * c8, c8], [3, 5, 3], 3),
            GResBlock(c8, c8, nc, 1), GResBlock(c8, c8, nc),
            ConvGRU(c8, [c8, 2 * c8, c8], [3, 5, 3], 3),
            GResBlock(c8, c8, nc, 1), GResBlock(c8, c8, nc),
            ConvGRU(c8, [c8, 2 * c8, c8], [3, 5, 3], 3),
            GResBlock(c8, c8, nc, 1), GResBlock(c8, c4, nc),
            ConvGRU(c4, [c4, 2 * c4, c4], [3, 5, 5], 3),
            GResBlock(c4, c4, nc, 1), GResBlock(c4, c2, nc),
        ])
        self.colorize = SpectralNormConv(c2, 3, (3, 3))
        if self_attn or sep_attn:
            from .attention3d import SelfAttention, SeparableAttn
            if self_attn:
                self.self_attn = SelfAttention(c8, compute_dtype)
            if sep_attn:
                self.sep_attn = SeparableAttn(c4, compute_dtype)
        self.dp_global = False                      # data-parallel "global" mode: conditions gathered over the ranks
        self.dp_hooks = False                       # data-parallel trainer sets it: stage-boundary gradient hooks
        self.grad_ready_hook = None                 # callable(first finished module index), armed around backward
        self.grad_ready_stages = (2, 5, 8)          # after each [ConvGRU, GResBlock, GResBlock] group but the last

    def forward(self, x, class_id, hidden=None):
        """hidden (frame-conditional variant, BASELINE configs[4]): initial ConvGRU states carried in from a conditioning
        encoder -- one entry per ConvGRU of the stack (4), each None or a list of that ConvGRU's per-layer states
        [B, hidden_l, S, S] fp32 (None entries = zeros).  They take the place of the `hidden=None` the reference passes at
        the first frame (Generator.py:91,96 -> ConvGRU.forward(x, hidden), ConvGRU.py:104-118) and receive gradients."""
        sn = prefetch_spectral_norm(self, self.compute_dtype)    # SN + weight packing of all layers on the side stream
        counted = self._count_batches() if self.training else []
        for m in counted:
            m.count_batches = False
        try:
            return self._forward(x, class_id, hidden)
        finally:
            for m in counted:                     # (only modules that were counting are switched back on)
                m.count_batches = True
            clear_spectral_norm(sn)

    def _count_batches(self):
        """BatchNorm2d's `num_batches_tracked += 1` of all sixteen conditional batch norms (Normalization.py:72, train mode) as ONE
        multi-tensor launch.  The counters stay the modules' own buffers (nothing is re-homed: shadow copies, EMA code and
        `.to()` keep seeing the tensors they hold); the layers' own increment is switched off only for the duration of this
        forward, so a ConditionalNorm / GResBlock driven on its own afterwards counts for itself again."""
        # like BatchNorm2d (Normalization.py:72) the counter follows each module's OWN training flag; a layer whose caller switched
        # its counting off (count_batches = False) stays untouched and keeps that setting
        mods = [m for m in self.modules() if isinstance(m, ConditionalNorm) and m.training and m.count_batches]
        if mods:
            torch._foreach_add_([m.bn.num_batches_tracked for m in mods], 1)
        return mods

    def _forward(self, x, class_id, hidden=None):
        B, T = x.shape[0], self.n_frames
        dev = x.device
        class_emb = Fn.Embedding.apply(self.embedding.weight, class_id.to(torch.int32))
        zc = torch.cat([x, class_emb], 1)
        y = Fn.LinearF32.apply(zc, self.affine_transfrom.weight, self.affine_transfrom.bias)
        y = y.view(B, 8 * self.ch, self.latent_dim, self.latent_dim)
        y = Fn.ToChannelsLast.apply(y, self.compute_dtype, None)
        # frame (t,b) is stored at t*B+b; the reference conditions frame b*T+t on row (b*T+t) mod B
        t_idx = torch.arange(T, device=dev).view(T, 1)
        b_idx = torch.arange(B, device=dev).view(1, B)
        samp = ((b_idx * T + t_idx) % B).reshape(-1).to(torch.int32)
        cond = zc
        if self.dp_global and D.exchange_on():
            # one process on the global batch would condition frame (b, t) on row (b*T + t) mod B_global: gather the
            # condition rows of all ranks (rank-major = global batch order) and index them with the GLOBAL b
            cond = D.AllGatherRows.apply(zc)
            b_glob = torch.distributed.get_rank() * B + b_idx
            samp = ((b_glob * T + t_idx) % cond.shape[0]).reshape(-1).to(torch.int32)
        n_gru = 0
        for k, m in enumerate(self.conv):
            if isinstance(m, ConvGRU):
                h0 = hidden[n_gru] if hidden is not None else None
                n_gru += 1
                if h0 is not None:
                    if len(h0) != m.n_layers:
                        raise ValueError("`hidden` needs one state (or None) per ConvGRU layer")
                    h0 = [None if h is None else Fn.ToChannelsLast.apply(h, self.compute_dtype, None) for h in h0]
                y = m.run(y, T, shared_x=(k == 0), hidden=h0)[-1]
            else:
                y = m.run(y, cond, samp)
            attn = getattr(self, "self_attn", None) if k == 0 else getattr(self, "sep_attn", None) if k == 8 else None
            if attn is not None:                      # clip-level attention: frames clip-major around the block
                S1, S2, Cp = y.shape[1:]
                clip = Fn.SwapFrameOrder.apply(y, T, B).view(B, T, S1, S2, Cp)
                y = Fn.SwapFrameOrder.apply(attn.run(clip).view(B * T, S1, S2, Cp), B, T)
            if self.dp_hooks and y.requires_grad and k in self.grad_ready_stages:
                # fires when the backward pass has produced d/dy: every module after k has all its gradients queued
                y.register_hook(lambda g_, k_=k: self.grad_ready_hook(k_ + 1) if self.grad_ready_hook else None)
        y = self.colorize(y, relu_in=True, act=L.ACT_TANH)
        out = Fn.FromChannelsLast.apply(y, 3, (B, T))          # t-major frames -> b-major [B*T,3,H,W]
        return out.view(B, T, 3, out.shape[-2], out.shape[-1])
