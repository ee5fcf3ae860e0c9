"""Spatio-temporal self attention of Module/Attention.py:114-185 (`SelfAttention`) on the HIP path.

The reference defines this block (and SeparableAttn) for the generator but never calls it (Generator.py has the calls
commented out); it is provided as an optional module with the reference's parameter names so a generator variant that
enables it can load the same state_dict.  Data flow, all tokens of a clip on one axis (N = T*W*H):
    q = Conv3d_1x1(x)  [N, C/2];   k = MaxPool3d(2)(Conv3d_1x1(x))  [N/8, C/2];   v = MaxPool3d(2)(Conv3d_1x1(x))  [N/8, C]
    y = gamma * softmax(q k^T) v + x
One fused 1x1 GEMM produces [q | k | v]; the pooled copy feeds the key / value side of the attention kernels.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import kern as K
from .disc_nets import QKVConv
from .sn_layers import PlainConv


class SelfAttention(nn.Module):
    """Attention.py:114-185.  Keys: gamma, {query,key,value}_conv.{weight [Cout,Cin,1,1,1], bias}.
    forward(x [B, C, T, W, H] fp32) -> same shape; T, W, H must be even (:161)."""

    def __init__(self, in_dim, compute_dtype=torch.bfloat16):
        super().__init__()
        self.chanel_in = in_dim
        self.compute_dtype = compute_dtype
        self.query_conv = PlainConv(in_dim, in_dim // 2, (1, 1, 1))
        self.key_conv = PlainConv(in_dim, in_dim // 2, (1, 1, 1))
        self.value_conv = PlainConv(in_dim, in_dim, (1, 1, 1))
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        if x.dim() != 5:
            raise RuntimeError("SelfAttention expects [B, C, T, W, H] (the reference's 4-D path fails its own assert)")
        B, C_, T, W, H = x.shape
        assert T % 2 == 0 and W % 2 == 0 and H % 2 == 0, "T, W, H is not even"
        xc = Fn.ToChannelsLast.apply(x, self.compute_dtype, None)            # [B, T, W, H, Cp]
        return Fn.FromChannelsLast.apply(self.run(xc), C_, None)

    def run(self, xc):
        """channels-last in / out"""
        C_, dq = self.chanel_in, self.chanel_in // 2
        dqp = K.pad8(dq)
        ctot = 2 * dqp + C_                                                  # q | pad | k | pad | v
        spec = Fn.ConvSpec((1, 1), ctot, C_)
        pk = K.PackedConv(xc.dtype, ctot, C_, (1, 1), xc.device)
        wq, wk, wv = (c.weight.view(c.cout, c.cin, 1, 1) for c in (self.query_conv, self.key_conv, self.value_conv))
        pk.fill(wq.data, co_off=0).fill(wk.data, co_off=dqp).fill(wv.data, co_off=2 * dqp)
        spec.pack = pk
        qkv = QKVConv.apply(xc, wq, self.query_conv.bias, wk, self.key_conv.bias, wv, self.value_conv.bias, spec, dq, dqp)
        kv = Fn.MaxPool3d.apply(qkv)                                         # the q columns ride along unused
        return Fn.SelfAttentionKV.apply(xc, qkv, kv, self.gamma, dq, C_)


class SeparableAttnCell(nn.Module):
    """Attention.py:24-111.  Keys: gamma, {query,key,value}_conv.{weight [Cout,Cin,1,1,1], bias}.  `attn_id` in 'T' | 'W' | 'H'."""

    def __init__(self, in_dim, attn_id=None, compute_dtype=torch.bfloat16):
        super().__init__()
        if attn_id not in ("T", "W", "H"):
            raise ValueError("attn_id must be 'T', 'W' or 'H'")
        self.attn_id, self.chanel_in, self.compute_dtype = attn_id, in_dim, compute_dtype
        self.query_conv = PlainConv(in_dim, in_dim // 2, (1, 1, 1))
        self.key_conv = PlainConv(in_dim, in_dim // 2, (1, 1, 1))
        self.value_conv = PlainConv(in_dim, in_dim, (1, 1, 1))
        self.gamma = nn.Parameter(torch.zeros((1,)))

    def forward(self, x):
        B, C_, T, W, H = x.shape
        xc = Fn.ToChannelsLast.apply(x, self.compute_dtype, None)
        return Fn.FromChannelsLast.apply(self.run(xc), C_, None)

    def run(self, xc):
        """channels-last [B, T, W, H, Cp] in / out"""
        B, T, W, H, _ = xc.shape
        assert T % 2 == 0 and W % 2 == 0 and H % 2 == 0, "T, W, H is not even"
        C_, dq = self.chanel_in, self.chanel_in // 2
        dqp = K.pad8(dq)
        ctot = 2 * dqp + C_
        spec = Fn.ConvSpec((1, 1), ctot, C_)
        pk = K.PackedConv(xc.dtype, ctot, C_, (1, 1), xc.device)
        wq, wk, wv = (c.weight.view(c.cout, c.cin, 1, 1) for c in (self.query_conv, self.key_conv, self.value_conv))
        pk.fill(wq.data, co_off=0).fill(wk.data, co_off=dqp).fill(wv.data, co_off=2 * dqp)
        spec.pack = pk
        qkv = QKVConv.apply(xc, wq, self.query_conv.bias, wk, self.key_conv.bias, wv, self.value_conv.bias, spec, dq, dqp)
        return Fn.SeparableAttnCellFn.apply(xc, qkv, self.gamma, dq, C_, "TWH".index(self.attn_id))


class SeparableAttn(nn.Module):
    """Attention.py:8-21: the T, W and H cells in sequence; state_dict keys `model.{0,1,2}.*`."""

    def __init__(self, in_dim, compute_dtype=torch.bfloat16):
        super().__init__()
        self.model = nn.Sequential(*(SeparableAttnCell(in_dim, a, compute_dtype) for a in "TWH"))

    def forward(self, x):
        xc = Fn.ToChannelsLast.apply(x, self.model[0].compute_dtype, None)
        return Fn.FromChannelsLast.apply(self.run(xc), x.shape[1], None)

    def run(self, xc):
        for cell in self.model:
            xc = cell.run(xc)
        return xc
