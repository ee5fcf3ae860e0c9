"""torch.autograd.Function wrappers: each forward / backward is a short sequence of C-ABI kernel
launches on the current HIP stream.  Tensors between them are channels-last device tensors in the
storage dtype (torch.float32 = exact mode, torch.bfloat16); parameters and their gradients are
fp32 in the reference layouts, so torch.optim / checkpoints see exactly the reference tensors.
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import kern as K
from . import lib as L


# ------------------------------------------------------------------ weight gradients on a side stream
# Weight gradients are leaves of the backward pass: nothing downstream of them runs before the optimizer.  When a
# parameter's .grad already exists as a persistent fp32 buffer (optim.FlatAdam keeps every .grad as a view of one flat
# buffer) the weight-gradient kernels -- and the spectral-norm backward that follows them -- accumulate STRAIGHT INTO
# that buffer on a second, lower-priority HIP stream instead of returning a fresh tensor through autograd:
#   * the ~700 zeros_like / accumulate launches per step of the autograd route disappear;
#   * the big MFMA-bound weight-gradient launches fill the CUs the serial part of the backward pass leaves idle (the
#     48-step BPTT of the 4x4 / 8x8 ConvGRUs runs 512-workgroup launches of 15-70 us back to back) and overlap the
#     HBM-bound CBN / gate kernels.
# The stream that ran backward() must wait for the side stream before the gradients are read.  That join is AUTOMATIC:
# the first side-stream launch of a backward pass queues `join_side` as a final callback of the autograd engine, so it
# runs when that backward() returns -- whoever called it (the Trainer, a drop-in loop with a stock optimizer and
# zero_grad(set_to_none=False), gradient accumulation, a tool that drives tr.G directly); it is queued once per backward().  The Trainer additionally joins
# early where a gradient bucket is handed to the exchange.  Parameters without a persistent .grad (stock optimizers with
# zero_grad(set_to_none=True), module tests) take the autograd route unchanged.
import os as _os
_SIDE = {"on": False, "stream": None,
         "serial": _os.environ.get("DVD_SIDE_SERIAL") == "1"}      # env: profiling aid (rocprofv3 runs), see serialize_weight_grads


def direct_weight_grads(flag):
    """Switch the side-stream / direct-accumulation route on or off (Trainer turns it on; DVD_SIDE_WGRAD=0 vetoes)."""
    _SIDE["on"] = bool(flag) and _os.environ.get("DVD_SIDE_WGRAD", "1") != "0"


def serialize_weight_grads(flag):
    """Measurement aid (bench.py's instrumented step): keep the direct accumulation but launch the weight-gradient kernels on
    the CURRENT stream, so every kernel has the GPU to itself while its duration is being taken."""
    _SIDE["serial"] = bool(flag)


def side_stream():
    if _SIDE["stream"] is None:
        lo = 0
        try:
            lo = torch.cuda.Stream.priority_range()[0]           # (least, greatest) priority; smaller = more urgent
        except Exception:
            pass
        _SIDE["stream"] = torch.cuda.Stream(priority=lo)
    return _SIDE["stream"]


def side_stream_if_any():
    """The weight-gradient stream if one has been created (None before the first side launch)."""
    return _SIDE["stream"]


def _side_run(fn, *tensors):
    """fn() launches weight-gradient kernels: on the side stream, ordered after the current stream's queue."""
    with _on_side(*tensors):
        fn()


def join_side():
    """The current stream waits for everything queued on the weight-gradient stream (idempotent, one event wait)."""
    if _SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])


def _queue_join():
    """Called from inside a backward pass: make the running backward() end with join_side().  Queued ONCE per backward(): the
    autograd engine numbers its graph tasks, and the id of the task that last queued the callback is remembered -- a backward
    that raised (the engine drops its callbacks) cannot disable the join of the next one, which runs under a new id.  (The
    first form queued the callback on every side-stream launch: hundreds of event record / wait pairs at the end of every
    backward pass.)"""
    task = torch._C._current_graph_task_id()
    if task < 0:                  # not inside a backward pass (a Function.backward driven by hand): the caller joins
        return
    if _SIDE.get("task") == task:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join_side)
        _SIDE["task"] = task
    except RuntimeError:
        pass


def _direct(*params):
    """True when every given parameter has a persistent fp32 .grad the kernels can accumulate into."""
    if not _SIDE["on"]:
        return False
    for p in params:
        if p is None:
            continue
        if not p.is_leaf:                 # (a detached view of a frozen weight: reading .grad of a non-leaf only draws a warning)
            return False
        g = getattr(p, "grad", None)
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or not p.requires_grad:
            return False
    return True


class _on_side:
    """with _on_side(t1, t2, ...): kernels launched inside run on the side stream, ordered after everything queued on
    the current stream so far; the listed tensors (allocated on the current stream) are kept alive for it."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        if _SIDE["serial"]:
            self.cm = None
            return torch.cuda.current_stream()
        side = side_stream()
        _queue_join()
        side.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            t.record_stream(side)
        self.cm = torch.cuda.stream(side)
        self.cm.__enter__()
        return side

    def __exit__(self, *exc):
        return self.cm.__exit__(*exc) if self.cm is not None else False


# ------------------------------------------------------------------ boundary layout
class ToChannelsLast(Function):
    """fp32 [F, C, *spatial] -> [F', *spatial, pad8(C)]   (swap=(A,B): frame (a,b) -> (b,a))"""

    @staticmethod
    def forward(ctx, x, dtype, swap):
        ctx.channels, ctx.swap = x.shape[1], swap
        return K.to_cl(x, dtype, swap)

    @staticmethod
    def backward(ctx, g):
        return K.from_cl(g.contiguous(), ctx.channels, ctx.swap), None, None


class FromChannelsLast(Function):
    @staticmethod
    def forward(ctx, t, channels, swap):
        ctx.dtype, ctx.swap = t.dtype, swap
        return K.from_cl(t, channels, swap)

    @staticmethod
    def backward(ctx, g):
        return K.to_cl(g.contiguous(), ctx.dtype, ctx.swap), None, None


class SwapFrameOrder(Function):
    """[A*B frames, ...] a-major -> b-major (a pure row permutation of a channels-last tensor; (T, B) t-major <-> (B, T)
    clip-major around the clip-level attention blocks).  Rows are moved as 32-bit words, whatever the storage type."""

    @staticmethod
    def _perm(A, B, dev):
        return (torch.arange(B, device=dev).view(B, 1) + torch.arange(A, device=dev).view(1, A) * B).reshape(-1).int()

    @staticmethod
    def _move(x, idx):
        F_ = x.shape[0]
        words = x.contiguous().view(F_, -1).view(torch.float32)
        out = K.row_copy(words, idx, F_, words.shape[1], scatter=False)
        return out.view(x.dtype).view(x.shape)

    @staticmethod
    def forward(ctx, x, A, B):
        ctx.AB = (A, B)
        return SwapFrameOrder._move(x, SwapFrameOrder._perm(A, B, x.device))       # out[b*A + a] = x[a*B + b]

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.AB
        return SwapFrameOrder._move(g.contiguous(), SwapFrameOrder._perm(B, A, g.device)), None, None


# ------------------------------------------------------------------ convolution
POOL2_FUSED = _os.environ.get("DVD_POOL2_FUSED", "1") != "0"      # A/B aid: 0 = up2 backward-data as conv + dvd_pool (round-5 form)


class ConvSpec:
    """Static description + per-forward state of one convolution call."""

    def __init__(self, ksize, cout, cin, *, act=L.ACT_NONE, up2=False, relu_in=False, sn=None):
        self.ksize, self.cout, self.cin = tuple(ksize), cout, cin
        self.act, self.up2, self.relu_in = act, up2, relu_in
        self.sn = sn            # (u, v) parameter tensors of a spectral-norm wrapper, or None
        self.sigma = None       # set per forward
        self.pack = None


class GradSlot:
    """Hand-over of one branch's input gradient to the other branch of a two-consumer tensor (x feeds the main branch AND the
    1x1 shortcut of a residual block, GResBlock.py:50,70-73): the main branch's backward leaves its dx here instead of returning
    it, the shortcut conv's backward adds it in the epilogue of its backward-data conv (`res`) and returns the SUM -- the
    autograd engine would otherwise add the two gradients with a three-pass elementwise kernel.  The shortcut conv returns a
    0-element token that the main branch takes as an input, so the engine cannot run the shortcut's backward first."""

    def __init__(self):
        self.dx = None


class Conv(Function):
    """y = act(conv(x; w / sigma) + b [+ res])        (spec.pack is filled by the caller)
    slot (a GradSlot): this conv is the shortcut branch -> returns (y, token) and adds slot.dx to its input gradient."""

    @staticmethod
    def forward(ctx, x, w, b, res, spec, slot=None):
        pk = spec.pack
        # a residual at half the output size is read through a nearest x2 upsample (the generator's 1x1 shortcut runs before
        # its upsample: GResBlock.py:72-73 commute exactly)
        ru = res is not None and res.shape[-2] * 2 == K._grid(x, spec.ksize, spec.up2)[3]
        y = K.conv_forward(x, pk.wf, spec.ksize, spec.cout, bias=b, res=res, act=spec.act, up2=spec.up2,
                           relu_in=spec.relu_in, res_up2=ru, wq=lambda: pk.fragment_major("wf"))
        ctx.spec, ctx.pk, ctx.sigma = spec, pk, spec.sigma
        ctx.has_res, ctx.res_up2 = res is not None, ru
        ctx.params = (w, b)                      # the Parameter objects themselves (their .grad buffers)
        ctx.save_for_backward(x, w, y if spec.act != L.ACT_NONE else None)
        ctx.slot = slot
        if slot is not None:
            return y, x.new_empty(0)
        return y

    @staticmethod
    def backward(ctx, dy, dtok=None):
        x, w, y = ctx.saved_tensors
        spec, pk = ctx.spec, ctx.pk
        dy = dy.contiguous()
        if spec.act != L.ACT_NONE:
            dy = K.act_backward(dy, y, spec.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if spec.up2:
                # transpose of nearest x2 (+ ReLU mask): 2 x 2 sums in the conv's epilogue (round 6), else a pool pass over the
                # full-size result (exact mode on the tap-by-tap kernels, frames below 16 pixels)
                dx = K.conv_forward(dy, pk.wd, spec.ksize, pk.cip, wq=lambda: pk.fragment_major("wd"), pool2=True,
                                    mask=x if spec.relu_in else None) if POOL2_FUSED else None
                if dx is None:
                    dx = K.conv_forward(dy, pk.wd, spec.ksize, pk.cip, wq=lambda: pk.fragment_major("wd"))
                    dx = K.pool(dx, 1, scale=1.0, mask=x if spec.relu_in else None)
            else:
                partner = None
                if ctx.slot is not None:                  # the other branch's gradient of the same x: summed in the epilogue
                    partner, ctx.slot.dx = ctx.slot.dx, None
                    assert not spec.relu_in
                    # partner None: the main branch's backward node did not run in this pass.  The token the main branch takes from
                    # this conv orders its node BEFORE this one whenever the engine runs it at all, so a missing partner means the
                    # engine pruned the main branch -- torch.autograd.grad(loss, [block.conv_sc.weight]) needs no d/dx of the block
                    # input, runs this node for its weight only and DISCARDS the dx computed here (needs_input_grad[0] is fixed at
                    # forward time).  Nothing is lost: dx is the shortcut's own term (tests/test_gpu_modules.py, pruned-graph case).
                dx = K.conv_forward(dy, pk.wd, spec.ksize, pk.cip, mask=x if spec.relu_in else None, res=partner,
                                    wq=lambda: pk.fragment_major("wd"))
        wp, bp = ctx.params
        if ctx.needs_input_grad[1] and _direct(wp, bp if ctx.needs_input_grad[2] else None):
            # side stream, straight into the persistent .grad buffers (see the note at the top of this file)
            dbp = bp.grad if (ctx.needs_input_grad[2] and bp is not None) else None
            sigma = ctx.sigma

            def wgrad():
                if spec.sn is not None:
                    G = torch.empty_like(w)               # written, not accumulated (`overwrite`): no zero fill
                    K.conv_wgrad(x, dy, G, spec.ksize, spec.cout, spec.cin, up2=spec.up2, relu_in=spec.relu_in, dbias=dbp,
                                 overwrite=True)
                    u, v = spec.sn        # CURRENT u / v on purpose (reference quirk 7): no forward runs before the join
                    K.sn_backward(G, w, u, v, sigma, out=wp.grad)
                else:
                    K.conv_wgrad(x, dy, wp.grad, spec.ksize, spec.cout, spec.cin, up2=spec.up2, relu_in=spec.relu_in, dbias=dbp)
            _side_run(wgrad, x, dy, w, sigma)
        elif ctx.needs_input_grad[1]:
            G = torch.zeros_like(w)
            if ctx.needs_input_grad[2]:           # bias gradient rides along in the wgrad kernel
                db = torch.zeros(spec.cout, dtype=torch.float32, device=dy.device)
            K.conv_wgrad(x, dy, G, spec.ksize, spec.cout, spec.cin, up2=spec.up2, relu_in=spec.relu_in, dbias=db)
            if spec.sn is not None:
                u, v = spec.sn            # CURRENT u / v on purpose (reference quirk 7)
                dw = K.sn_backward(G, w, u, v, ctx.sigma)
            else:
                dw = G
        elif ctx.needs_input_grad[2]:
            db = K.colsum(dy, spec.cout)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = K.pool(dy, 1, scale=1.0) if ctx.res_up2 else dy          # transpose of nearest x2: 2 x 2 sums
        return dx, dw, db, dres, None, None


class Pool(Function):
    """average pool over (pt,2,2)"""

    @staticmethod
    def forward(ctx, x, pt):
        ctx.pt, ctx.hw = pt, (x.shape[-3], x.shape[-2])
        return K.pool(x, pt)

    @staticmethod
    def backward(ctx, g):
        return K.unpool(g.contiguous(), ctx.pt, out_hw=ctx.hw), None


# ------------------------------------------------------------------ fp32 linear / embedding
class LinearF32(Function):
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        ctx.params = (W, b)
        return K.linear_forward(x.contiguous(), W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        Wp, bp = ctx.params
        need_in, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_w and _direct(Wp, bp if ctx.has_bias else None):
            # persistent .grad buffers: the kernels accumulate straight into them (no zero fill, no autograd add per parameter)
            din = K.linear_backward(g.contiguous(), x.contiguous(), W, need_in, True, ctx.has_bias,
                                    dW=Wp.grad, db=bp.grad if ctx.has_bias else None)[0]
            return din, None, None
        din, dW, db = K.linear_backward(g.contiguous(), x.contiguous(), W, need_in, need_w, ctx.has_bias)
        return din, dW, db


class Embedding(Function):
    @staticmethod
    def forward(ctx, W, idx32):
        ctx.save_for_backward(idx32)
        ctx.rows = W.shape[0]
        ctx.param = W
        return K.row_copy(W, idx32, idx32.numel(), W.shape[1], scatter=False)

    @staticmethod
    def backward(ctx, g):
        (idx32,) = ctx.saved_tensors
        if _direct(ctx.param):
            K.embedding_backward(g.contiguous(), idx32, ctx.rows, dW=ctx.param.grad)
            return None, None
        return K.embedding_backward(g.contiguous(), idx32, ctx.rows), None


# ------------------------------------------------------------------ conditional batch norm (+ReLU)
class CondBatchNorm(Function):
    """y = relu?(gb[s][:C] * bn(x) + gb[s][C:]),  s = samp[frame]     Normalization.py:78-88"""

    @staticmethod
    def forward(ctx, x, gb, samp, C_real, relu, training, run_mean, run_var, eps, momentum, replicas=None, sums=None, tok=None,
                slot=None):
        """replicas: None (per-replica statistics = nn.DataParallel, trainer.py:357) or (world, all_reduce_sum_) for
        cross-replica batch norm (Generator.py:57 TODO): statistics and their backward sums span the global batch.
        sums: the module's persistent zeroed statistics workspace (kern.bn_stats)."""
        mean, rstd = K.bn_stats(x, C_real, training, eps, momentum, run_mean, run_var, replicas if training else None, sums)
        y = K.cbn_apply(x, C_real, mean, rstd, gb, samp, relu)
        ctx.save_for_backward(x, gb, samp, mean, rstd)      # (not y: the backward re-evaluates the ReLU mask from x)
        ctx.C_real, ctx.relu, ctx.training, ctx.replicas = C_real, relu, training, replicas
        ctx.slot = slot if tok is not None else None       # (tok: the shortcut conv's token, see GradSlot)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gb, samp, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("CondBatchNorm backward is implemented for training mode only")
        dx, dgb = K.cbn_backward(g.contiguous(), None, x, ctx.C_real, mean, rstd, gb, samp, ctx.relu, ctx.replicas)
        if ctx.slot is not None and ctx.needs_input_grad[0]:
            ctx.slot.dx, dx = dx, None                     # picked up (and added) by the shortcut conv's backward
        return dx, dgb, None, None, None, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------ ConvGRU layer
def _gru_gate_wgrads(x, dgx, dg, h_all, hr_all, h0, params, meta, weights):
    """Weight / bias gradients of the three gates of one ConvGRU layer, batched over all T steps (autograd of ConvGRU.py:49-51).
    x: the layer's input frames, dgx: d(pre-activation) as seen by the x-part (dg, or its sum over T for a shared input),
    dg: [T*B,S,S,3h].  Returns (grads[3], dbias[3]): None entries when the kernels accumulated straight into the persistent
    .grad buffers on the side stream."""
    T, B, S1, S2, hid, cin, k = meta
    dev = x.device
    ctot = cin + hid
    hflat = h_all.view(T * B, S1, S2, hid)
    hrflat = hr_all.view(T * B, S1, S2, hid)
    wparams = params[0::2]
    bparams = params[1::2]
    direct = _direct(*params)

    def gate_wgrads(dws, dbs):
        """dws[g] / dbs[g]: fp32 buffers (reference layouts) the kernels ACCUMULATE into"""
        for g in range(3):
            dw = dws[g]
            K.conv_wgrad(x, dgx, dw, (k, k), hid, cin, dy_col=g * hid, dw_ci_off=0, dw_ci_tot=ctot, dbias=dbs[g])
            if h0 is not None:
                # step 0 read the supplied state: h0 (update / reset) or h0 * r_0 = hr_all[0] (out gate)
                K.conv_wgrad(h0 if g < 2 else hrflat, dg, dw, (k, k), hid, hid, dy_col=g * hid, dw_ci_off=cin,
                             dw_ci_tot=ctot, frames=B, x_row0=0, dy_row0=0)
            if T > 1:
                # h-part: steps 1..T-1 read h_{t-1} (update/reset) or h_{t-1}*r_t (out gate)
                if g < 2:
                    K.conv_wgrad(hflat, dg, dw, (k, k), hid, hid, dy_col=g * hid, dw_ci_off=cin, dw_ci_tot=ctot,
                                 frames=(T - 1) * B, x_row0=0, dy_row0=B)
                else:
                    K.conv_wgrad(hrflat, dg, dw, (k, k), hid, hid, dy_col=g * hid, dw_ci_off=cin, dw_ci_tot=ctot,
                                 frames=(T - 1) * B, x_row0=B, dy_row0=B)

    if direct:
        _side_run(lambda: gate_wgrads([p.grad for p in wparams], [p.grad for p in bparams]), x, dgx, dg, h_all, hr_all, h0)
        return [None, None, None], [None, None, None]
    db3 = torch.zeros(3 * hid, dtype=torch.float32, device=dev)
    grads = [torch.zeros_like(w) for w in weights]
    gate_wgrads(grads, [db3[g * hid:(g + 1) * hid] for g in range(3)])
    return grads, [db3[:hid].clone(), db3[hid:2 * hid].clone(), db3[2 * hid:].clone()]


GRU_STACK = _os.environ.get("DVD_GRU_STACK", "1") != "0"      # A/B aid: 0 = every ConvGRU layer by layer (ConvGRULayer)
GRU_STACK_LAYER_POLICY = 0   # dvd_gru_stack_desc.layer_policy (tests set 1: the per-layer path's split-K factors -> bit-equal results)
GRU_NS_CAP = 0               # dvd_gru_desc.ns_cap (tests: a common upper bound on the split-K factors of both paths)
GRU_COMBINE_MAX = 0          # dvd_gru_desc.combine_max: 0 = the library's default policy (tests raise it to cover every slice count)


class ConvGRULayer(Function):
    """All T steps of one ConvGRUCell (ConvGRU.py:29-54).  x: [T*B,S,S,Cin_p] t-major frames, or
    [B,S,S,Cin_p] when the same input feeds every step (first GRU of the generator).
    Returns h for every step as [T*B,S,S,hidden]."""

    @staticmethod
    def forward(ctx, x, wu, bu, wr, br, wo, bo, T, shared_x, h0, infer=False):
        """infer (the caller runs under torch.no_grad(): the sampling path, trainer.py:323-334): nothing is kept for a
        backward pass -- u and h*r live in one-step scratch buffers, r and o are not stored at all (4 of the 5 [T, ...] tensors
        of the training forward are never allocated, a seventh of the gate epilogues' traffic is not written)."""
        dev, dtype = x.device, x.dtype
        hid, ctot, k = wu.shape[0], wu.shape[1], wu.shape[-1]
        cin = ctot - hid
        B = x.shape[0] if shared_x else x.shape[0] // T
        S1, S2 = x.shape[1], x.shape[2]
        M = B * S1 * S2
        px = K.PackedConv(dtype, 3 * hid, cin, (k, k), dev, covered=True)     # the fills below write every output channel
        pur = K.PackedConv(dtype, 2 * hid, hid, (k, k), dev, covered=True)
        po = K.PackedConv(dtype, hid, hid, (k, k), dev, covered=True)
        for g, w in enumerate((wu, wr, wo)):
            px.fill(w, co_off=g * hid, ci_off=0)
        pur.fill(wu, co_off=0, ci_off=cin).fill(wr, co_off=hid, ci_off=cin)
        po.fill(wo, ci_off=cin)
        bias3 = torch.cat([bu, br, bo])
        gx = K.conv_forward(x, px.wf, (k, k), 3 * hid, bias=bias3, wq=lambda: px.fragment_major("wf"))
        mk = lambda n=T: torch.empty(n, B, S1, S2, hid, dtype=dtype, device=dev)
        if infer:
            h_all, u_all, hr_all, r_all, o_all = mk(), mk(1), mk(1), None, None
        else:
            h_all, u_all, r_all, o_all, hr_all = mk(), mk(), mk(), mk(), mk()
        h32 = torch.empty(2, M, hid, dtype=torch.float32, device=dev) if dtype != torch.float32 else None
        lib = L.lib()
        ntaps = k * k
        ns1 = lib.dvd_conv_pick_nsplit(L.dt(x), C.c_longlong(M), 2 * hid, hid, ntaps)
        ns2 = lib.dvd_conv_pick_nsplit(L.dt(x), C.c_longlong(M), hid, hid, ntaps)
        ns3 = lib.dvd_conv_pick_nsplit(L.dt(x), C.c_longlong(M), hid, 2 * hid, ntaps)
        if GRU_NS_CAP > 0:
            ns1, ns2, ns3 = min(ns1, GRU_NS_CAP), min(ns2, GRU_NS_CAP), min(ns3, GRU_NS_CAP)
        ws_n = lib.dvd_convgru_ws_floats(L.dt(x), B, S1, S2, hid, k)
        ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
        d = L.GruDesc()
        d.dtype, d.T, d.B, d.H, d.W, d.hidden, d.k = L.dt(x), T, B, S1, S2, hid, k
        d.gx_stride = 0 if shared_x else M * 3 * hid
        d.gx, d.w_ur, d.w_o = gx.data_ptr(), pur.wf.data_ptr(), po.wf.data_ptr()
        # the recurrent convolutions on frames of 16 pixels and more read their weights in fragment-major order
        if K.wants_fragment_major(dtype, B, S1, S2, hid, 2 * hid, k, ns1):
            d.w_ur_q = pur.fragment_major("wf").data_ptr()
        if K.wants_fragment_major(dtype, B, S1, S2, hid, hid, k, ns2):
            d.w_o_q = po.fragment_major("wf").data_ptr()
        d.h0 = h0.data_ptr() if h0 is not None else None
        d.h_all, d.u_all, d.hr_all = h_all.data_ptr(), u_all.data_ptr(), hr_all.data_ptr()
        d.r_all = r_all.data_ptr() if r_all is not None else None
        d.o_all = o_all.data_ptr() if o_all is not None else None
        d.h32 = h32.data_ptr() if h32 is not None else None
        d.ws = ws.data_ptr()
        d.tickets = L.gru_tickets(dev).data_ptr()
        d.combine_max = GRU_COMBINE_MAX
        d.ns_cap = GRU_NS_CAP
        d.infer = int(infer)
        L.check(lib.dvd_convgru_layer_forward(C.byref(d), L.stream()))
        if infer:
            return h_all.view(T * B, S1, S2, hid)
        ctx.save_for_backward(x, wu, wr, wo, h_all, u_all, r_all, o_all, hr_all, h0)
        ctx.params = (wu, bu, wr, br, wo, bo)
        ctx.packs = (px, pur, po)
        ctx.meta = (T, B, S1, S2, hid, cin, k, shared_x, ws_n)
        return h_all.view(T * B, S1, S2, hid)

    @staticmethod
    def backward(ctx, dh):
        x, wu, wr, wo, h_all, u_all, r_all, o_all, hr_all, h0 = ctx.saved_tensors
        px, pur, po = ctx.packs
        T, B, S1, S2, hid, cin, k, shared_x, ws_n = ctx.meta
        dev, dtype = x.device, x.dtype
        M = B * S1 * S2
        dh = dh.contiguous()
        dg = torch.empty(T * B, S1, S2, 3 * hid, dtype=dtype, device=dev)
        carry = torch.empty(M, hid, dtype=torch.float32, device=dev)
        ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
        d = L.GruDesc()
        d.dtype, d.T, d.B, d.H, d.W, d.hidden, d.k = L.dt(x), T, B, S1, S2, hid, k
        d.wd_ur, d.wd_o = pur.wd.data_ptr(), po.wd.data_ptr()
        lib = L.lib()
        ns_o = lib.dvd_conv_pick_nsplit(L.dt(x), C.c_longlong(M), hid, hid, k * k)
        ns_ur = lib.dvd_conv_pick_nsplit(L.dt(x), C.c_longlong(M), hid, 2 * hid, k * k)
        if GRU_NS_CAP > 0:
            ns_o, ns_ur = min(ns_o, GRU_NS_CAP), min(ns_ur, GRU_NS_CAP)
        if K.wants_fragment_major(dtype, B, S1, S2, 2 * hid, hid, k, ns_ur):
            d.wd_ur_q = pur.fragment_major("wd").data_ptr()
        if K.wants_fragment_major(dtype, B, S1, S2, hid, hid, k, ns_o):
            d.wd_o_q = po.fragment_major("wd").data_ptr()
        d.h0 = h0.data_ptr() if h0 is not None else None
        d.h_all, d.u_all, d.r_all = h_all.data_ptr(), u_all.data_ptr(), r_all.data_ptr()
        d.o_all, d.hr_all = o_all.data_ptr(), hr_all.data_ptr()
        d.ws, d.dh_out, d.dg, d.carry = ws.data_ptr(), dh.data_ptr(), dg.data_ptr(), carry.data_ptr()
        d.tickets = L.gru_tickets(dev).data_ptr()
        d.combine_max = GRU_COMBINE_MAX
        d.ns_cap = GRU_NS_CAP
        dh0_32 = None
        if h0 is not None and ctx.needs_input_grad[9]:          # gradient wrt the supplied initial state (ConvGRU.py:104)
            dh0_32 = torch.empty(M, hid, dtype=torch.float32, device=dev)
            d.dh0 = dh0_32.data_ptr()
        L.check(L.lib().dvd_convgru_layer_backward(C.byref(d), L.stream()))
        # ---- everything below is batched over all T steps ----
        dgx = K.sum_leading(dg.view(T, -1)).view(B, S1, S2, 3 * hid) if shared_x else dg
        dx = K.conv_forward(dgx, px.wd, (k, k), px.cip, wq=lambda: px.fragment_major("wd")) if ctx.needs_input_grad[0] else None
        grads, dbl = _gru_gate_wgrads(x, dgx, dg, h_all, hr_all, h0, ctx.params, (T, B, S1, S2, hid, cin, k), (wu, wr, wo))
        dh0 = None
        if dh0_32 is not None:
            dh0 = dh0_32.view(h0.shape) if h0.dtype == torch.float32 else K.convert(dh0_32, h0.dtype).view(h0.shape)
        return (dx, grads[0], dbl[0], grads[1], dbl[1], grads[2], dbl[2], None, None, dh0, None)


class ConvGRUStack(Function):
    """All layers and all T steps of a ConvGRU (ConvGRU.py:57-133) as a LAYER WAVEFRONT (dvd_convgru_stack_*, gru.hip): layer l
    works on step t while layer l-1 works on step t + 2, the independent recurrent / x-part convolutions of the layers run as
    grouped launches.  Arguments: x, T, shared_x, infer, n_layers, then per layer (wu, bu, wr, br, wo, bo), then per layer h0
    (or None).  Returns the per-layer state sequences [T*B,S,S,h_l].  Same results as ConvGRULayer applied layer by layer."""

    @staticmethod
    def usable(x, cells):
        """Host-side pre-check (the library's dvd_convgru_stack_ok has the last word inside forward)."""
        if x.dtype != torch.bfloat16 or not GRU_STACK or len(cells) > L.GRU_STACK_MAX:
            return False
        S1, S2 = x.shape[1], x.shape[2]
        if S1 != S2 or S1 & (S1 - 1) or not (S1 in (4, 8) or S1 >= 16):
            return False
        return all(c.kernel_size in (3, 5) and c.hidden_size % 8 == 0 for c in cells)

    @staticmethod
    def forward(ctx, x, T, shared_x, infer, nl, *flat):
        dev, dtype = x.device, x.dtype
        lib = L.lib()
        wts = [flat[6 * l:6 * l + 6] for l in range(nl)]
        h0s = list(flat[6 * nl:7 * nl])
        B = x.shape[0] if shared_x else x.shape[0] // T
        S1, S2 = x.shape[1], x.shape[2]
        M = B * S1 * S2
        sd = L.GruStackDesc()
        sd.n_layers, sd.layer_policy, sd.run = nl, GRU_STACK_LAYER_POLICY, 0
        keep, layers = [], []
        inp = x
        # the 6 gate packs of every layer and their fragment-major images (the backward pass's too unless `infer`) in TWO launches for
        # the whole ConvGRU (K.PackBatch) instead of 12-15 per layer, all of them in front of the first convolution
        pb, packs = K.PackBatch(), []
        for l, (wu, bu, wr, br, wo, bo) in enumerate(wts):
            hid, ctot, k = wu.shape[0], wu.shape[1], wu.shape[-1]
            cin = ctot - hid
            px = K.PackedConv(dtype, 3 * hid, cin, (k, k), dev, covered=True)
            pur = K.PackedConv(dtype, 2 * hid, hid, (k, k), dev, covered=True)
            po = K.PackedConv(dtype, hid, hid, (k, k), dev, covered=True)
            for g, w in enumerate((wu, wr, wo)):
                pb.fill(px, w, co_off=g * hid, ci_off=0)
            pb.fill(pur, wu, co_off=0, ci_off=cin).fill(pur, wr, co_off=hid, ci_off=cin)
            pb.fill(po, wo, ci_off=cin)
            for pk in (px, pur, po):
                pb.fragment_major(pk, "wf")
                if not infer and (pk is not px or l > 0 or ctx.needs_input_grad[0]):
                    pb.fragment_major(pk, "wd")
            packs.append((px, pur, po))
        pb.run()
        for l, (wu, bu, wr, br, wo, bo) in enumerate(wts):
            hid, ctot, k = wu.shape[0], wu.shape[1], wu.shape[-1]
            cin = ctot - hid
            px, pur, po = packs[l]
            bias3 = torch.cat([bu, br, bo])
            if l == 0:      # the first layer's input is known for every step: one batched convolution, off the chain
                gx = K.conv_forward(x, px.wf, (k, k), 3 * hid, bias=bias3, wq=lambda: px.fragment_major("wf"))
            else:           # written step by step inside the wavefront
                gx = torch.empty(T * B, S1, S2, 3 * hid, dtype=dtype, device=dev)
                sd.cin[l] = K.pad8(cin)
                sd.wx[l], sd.wx_q[l], sd.bx[l] = px.wf.data_ptr(), px.fragment_major("wf").data_ptr(), bias3.data_ptr()
            mk = lambda n=T: torch.empty(n, B, S1, S2, hid, dtype=dtype, device=dev)
            if infer:
                h_all, u_all, hr_all, r_all, o_all = mk(), mk(1), mk(1), None, None
            else:
                h_all, u_all, r_all, o_all, hr_all = mk(), mk(), mk(), mk(), mk()
            h32 = torch.empty(2, M, hid, dtype=torch.float32, device=dev)
            h0 = h0s[l]
            d = sd.layer[l]
            d.dtype, d.T, d.B, d.H, d.W, d.hidden, d.k = L.dt(x), T, B, S1, S2, hid, k
            d.gx_stride = 0 if (shared_x and l == 0) else M * 3 * hid
            d.gx, d.w_ur, d.w_o = gx.data_ptr(), pur.wf.data_ptr(), po.wf.data_ptr()
            d.w_ur_q, d.w_o_q = pur.fragment_major("wf").data_ptr(), po.fragment_major("wf").data_ptr()
            d.h0 = h0.data_ptr() if h0 is not None else None
            d.h_all, d.u_all, d.hr_all = h_all.data_ptr(), u_all.data_ptr(), hr_all.data_ptr()
            d.r_all = r_all.data_ptr() if r_all is not None else None
            d.o_all = o_all.data_ptr() if o_all is not None else None
            d.h32 = h32.data_ptr()
            d.tickets = L.gru_tickets(dev).data_ptr()
            d.ns_cap = GRU_NS_CAP
            d.infer = int(infer)
            keep.append((gx, h32, bias3))
            layers.append(dict(px=px, pur=pur, po=po, h_all=h_all, u_all=u_all, r_all=r_all, o_all=o_all, hr_all=hr_all,
                               hid=hid, cin=cin, k=k))
        ws = torch.empty(lib.dvd_convgru_stack_ws_floats(C.byref(sd)), dtype=torch.float32, device=dev)
        sd.ws = ws.data_ptr()
        if not lib.dvd_convgru_stack_ok(C.byref(sd), 0):
            raise RuntimeError("ConvGRUStack: this stack is not served by the wavefront path (ConvGRUStack.usable disagrees with the library)")
        L.check(lib.dvd_convgru_stack_forward(C.byref(sd), L.stream()))
        outs = tuple(ly["h_all"].view(T * B, S1, S2, ly["hid"]) for ly in layers)
        if infer:
            return outs
        ctx.set_materialize_grads(False)          # unused layer outputs arrive as None, not as zero tensors
        saved = [x]
        for l, ly in enumerate(layers):
            saved += [wts[l][0], wts[l][2], wts[l][4], ly["h_all"], ly["u_all"], ly["r_all"], ly["o_all"], ly["hr_all"], h0s[l]]
        ctx.save_for_backward(*saved)
        ctx.params = [tuple(w) for w in wts]
        ctx.packs = [(ly["px"], ly["pur"], ly["po"]) for ly in layers]
        ctx.meta = (T, B, S1, S2, shared_x, nl, [(ly["hid"], ly["cin"], ly["k"]) for ly in layers])
        return outs

    @staticmethod
    def backward(ctx, *dhs):
        T, B, S1, S2, shared_x, nl, dims = ctx.meta
        sv = ctx.saved_tensors
        x = sv[0]
        per = [sv[1 + 9 * l:1 + 9 * (l + 1)] for l in range(nl)]
        dev, dtype = x.device, x.dtype
        lib = L.lib()
        M = B * S1 * S2
        sd = L.GruStackDesc()
        sd.n_layers, sd.layer_policy, sd.run = nl, GRU_STACK_LAYER_POLICY, 0
        dgs, carries, dh_mid, dh0_32, keep = [], [], [None] * nl, [None] * nl, []
        for l in range(nl):
            wu, wr, wo, h_all, u_all, r_all, o_all, hr_all, h0 = per[l]
            hid, cin, k = dims[l]
            px, pur, po = ctx.packs[l]
            dg = torch.empty(T * B, S1, S2, 3 * hid, dtype=dtype, device=dev)
            carry = torch.empty(M, hid, dtype=torch.float32, device=dev)
            d = sd.layer[l]
            d.dtype, d.T, d.B, d.H, d.W, d.hidden, d.k = L.dt(x), T, B, S1, S2, hid, k
            d.gx_stride = M * 3 * hid
            d.gx = dg.data_ptr()                         # (not read by the backward pass; the descriptor check wants it non-null)
            d.wd_ur, d.wd_o = pur.wd.data_ptr(), po.wd.data_ptr()
            d.wd_ur_q, d.wd_o_q = pur.fragment_major("wd").data_ptr(), po.fragment_major("wd").data_ptr()
            d.h0 = h0.data_ptr() if h0 is not None else None
            d.h_all, d.u_all, d.r_all = h_all.data_ptr(), u_all.data_ptr(), r_all.data_ptr()
            d.o_all, d.hr_all = o_all.data_ptr(), hr_all.data_ptr()
            d.dg, d.carry = dg.data_ptr(), carry.data_ptr()
            dh = dhs[l].contiguous() if dhs[l] is not None else None
            d.dh_out = dh.data_ptr() if dh is not None else None
            keep.append(dh)
            d.tickets = L.gru_tickets(dev).data_ptr()
            d.ns_cap = GRU_NS_CAP
            if l > 0:
                sd.cin[l] = K.pad8(cin)
                sd.wdx[l], sd.wdx_q[l] = px.wd.data_ptr(), px.fragment_major("wd").data_ptr()
                dh_mid[l] = torch.empty(T * B, S1, S2, K.pad8(cin), dtype=dtype, device=dev)
                sd.dh_mid[l] = dh_mid[l].data_ptr()
            if h0 is not None and ctx.needs_input_grad[5 + 6 * nl + l]:
                dh0_32[l] = torch.empty(M, hid, dtype=torch.float32, device=dev)
                d.dh0 = dh0_32[l].data_ptr()
            dgs.append(dg)
            carries.append(carry)
        ws = torch.empty(lib.dvd_convgru_stack_ws_floats(C.byref(sd)), dtype=torch.float32, device=dev)
        sd.ws = ws.data_ptr()
        L.check(lib.dvd_convgru_stack_backward(C.byref(sd), L.stream()))
        # ---- everything below is batched over all T steps ----
        out = []
        dx = None
        for l in range(nl):
            wu, wr, wo, h_all, u_all, r_all, o_all, hr_all, h0 = per[l]
            hid, cin, k = dims[l]
            px = ctx.packs[l][0]
            dg = dgs[l]
            if l == 0:
                dgx = K.sum_leading(dg.view(T, -1)).view(B, S1, S2, 3 * hid) if shared_x else dg
                if ctx.needs_input_grad[0]:
                    dx = K.conv_forward(dgx, px.wd, (k, k), px.cip, wq=lambda: px.fragment_major("wd"))
                xin = x
            else:
                dgx = dg
                xin = per[l - 1][3].view(T * B, S1, S2, dims[l - 1][0])          # the layer below's states
            grads, dbl = _gru_gate_wgrads(xin, dgx, dg, h_all, hr_all, h0, ctx.params[l], (T, B, S1, S2, hid, cin, k), (wu, wr, wo))
            out += [grads[0], dbl[0], grads[1], dbl[1], grads[2], dbl[2]]
            # this layer's d(pre-activations) are done with once its weight-gradient launches are queued (the side stream holds them
            # through record_stream): release them layer by layer instead of keeping every layer's [T*B,S,S,3h] until the return
            dg = dgx = None
            dgs[l] = carries[l] = dh_mid[l] = None
        dh0s = []
        for l in range(nl):
            h0 = per[l][8]
            if dh0_32[l] is None:
                dh0s.append(None)
            else:
                dh0s.append(dh0_32[l].view(h0.shape) if h0.dtype == torch.float32 else K.convert(dh0_32[l], h0.dtype).view(h0.shape))
        return (dx, None, None, None, None, *out, *dh0s)


# ------------------------------------------------------------------ attention / head / loss
ATTN_MFMA = _os.environ.get("DVD_ATTN_MFMA", "1") != "0"      # A/B aid: 0 = the fp32 vector-pipe attention kernels also in bf16 mode


class SelfAttention2d(Function):
    """y = gamma * softmax(q^T k) v + x      Discriminators.py:100-119; qkv from one fused 1x1 conv.
    bf16 storage at the discriminators' widths (16 query channels, 128 value channels, any token count that is a multiple of 32 up to
    4096 per frame -- dvd_attention_mfma_ok has the exact rule) runs
    on the matrix cores and keeps one number per query (dvd_attention_mfma_*); everything else runs the fp32 kernels, which keep
    the N x N map for the backward pass."""

    @staticmethod
    def forward(ctx, x, qkv, gamma, dq, C_real):
        F_ = x.shape[0]
        N = x.shape[1] * x.shape[2]
        koff = K.pad8(dq)
        voff = 2 * koff
        ctx.gamma_param = gamma
        lib = L.lib()
        ctx.mfma = bool(ATTN_MFMA and lib.dvd_attention_mfma_ok(L.dt(x), qkv.shape[-1], dq, koff, voff, x.shape[-1], C_real, N))
        ctx.meta = (dq, C_real, koff, voff, N)
        if ctx.mfma:
            y, att = torch.empty_like(x), torch.empty_like(x)
            lse = torch.empty(F_, N, dtype=torch.float32, device=x.device)
            L.check(lib.dvd_attention_mfma_forward(L.ptr(qkv), qkv.shape[-1], L.ptr(x), C_real, L.ptr(gamma), L.ptr(y),
                                                   L.ptr(att), L.ptr(lse), C.c_longlong(F_), N, L.stream()))
            ctx.save_for_backward(qkv, gamma, att, lse)
            return y
        y = torch.zeros_like(x) if x.shape[-1] != C_real else torch.empty_like(x)
        att = torch.zeros_like(x)
        A = torch.empty(F_, N, N, dtype=torch.float32, device=x.device)
        L.check(lib.dvd_attention_forward(L.dt(x), L.ptr(qkv), qkv.shape[-1], dq, koff, voff, L.ptr(x), x.shape[-1],
                                          C_real, L.ptr(gamma), L.ptr(y), L.ptr(att), L.ptr(A), C.c_longlong(F_), N,
                                          L.stream()))
        ctx.save_for_backward(qkv, gamma, att, A)
        return y

    @staticmethod
    def backward(ctx, dy):
        qkv, gamma, att, A = ctx.saved_tensors
        dq, C_real, koff, voff, N = ctx.meta
        dy = dy.contiguous()
        F_ = dy.shape[0]
        direct = ctx.needs_input_grad[2] and _direct(ctx.gamma_param)       # the kernel ADDS into dgamma: persistent .grad taken as is
        dgamma = ctx.gamma_param.grad if direct else torch.zeros(1, dtype=torch.float32, device=dy.device)
        if ctx.mfma:                                                         # (A is the [F, N] log-sum-exp here)
            dqkv = torch.empty_like(qkv) if qkv.shape[-1] == 32 + C_real else torch.zeros_like(qkv)
            D = torch.empty_like(A)
            L.check(L.lib().dvd_attention_mfma_backward(L.ptr(qkv), qkv.shape[-1], L.ptr(dy), C_real, L.ptr(gamma), L.ptr(att),
                                                        L.ptr(A), L.ptr(D), L.ptr(dqkv), L.ptr(dgamma), C.c_longlong(F_), N,
                                                        L.stream()))
            return dy, dqkv, (None if direct else dgamma), None, None
        dqkv = torch.zeros_like(qkv)
        dS = torch.empty_like(A)
        L.check(L.lib().dvd_attention_backward(L.dt(dy), L.ptr(qkv), qkv.shape[-1], dq, koff, voff, L.ptr(dy),
                                               dy.shape[-1], C_real, L.ptr(gamma), L.ptr(att), L.ptr(A), L.ptr(dS),
                                               L.ptr(dqkv), L.ptr(dgamma), C.c_longlong(F_), N, L.stream()))
        return dy, dqkv, (None if direct else dgamma), None, None


class MaxPool3d(Function):
    """nn.MaxPool3d(2, 2) on a channels-last [F, T, H, W, C] tensor (Attention.py:148)."""

    @staticmethod
    def forward(ctx, x):
        F_, T, H, W, ld = x.shape
        y = torch.empty(F_, T // 2, H // 2, W // 2, ld, dtype=x.dtype, device=x.device)
        L.check(L.lib().dvd_maxpool3d(L.dt(x), L.ptr(x), L.ptr(y), C.c_longlong(F_), T // 2, H // 2, W // 2, ld, L.stream()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        F_, T, H, W, ld = x.shape
        dx = torch.empty_like(x)
        L.check(L.lib().dvd_maxpool3d_backward(L.dt(x), L.ptr(x), L.ptr(dy), L.ptr(dx), C.c_longlong(F_), T // 2, H // 2,
                                               W // 2, ld, L.stream()))
        return dx


class SelfAttentionKV(Function):
    """y = gamma * softmax(q k^T) v + x with the keys / values on their own (pooled) token set
    (Attention.py:160-179).  q: columns [0, dq) of `qkv` (N tokens per clip); k, v: columns [koff, koff+dq) and
    [voff, voff+C) of `kv` (Nk tokens per clip)."""

    @staticmethod
    def forward(ctx, x, qkv, kv, gamma, dq, C_real):
        F_ = x.shape[0]
        N = x.numel() // (F_ * x.shape[-1])
        Nk = kv.numel() // (F_ * kv.shape[-1])
        koff = K.pad8(dq)
        voff = 2 * koff
        y = torch.zeros_like(x) if x.shape[-1] != C_real else torch.empty_like(x)
        att = torch.zeros_like(x)
        A = torch.empty(F_, N, Nk, dtype=torch.float32, device=x.device)
        L.check(L.lib().dvd_attention_kv_forward(L.dt(x), L.ptr(qkv), qkv.shape[-1], dq, L.ptr(kv), kv.shape[-1], koff, voff,
                                                 L.ptr(x), x.shape[-1], C_real, L.ptr(gamma), L.ptr(y), L.ptr(att), L.ptr(A),
                                                 C.c_longlong(F_), N, Nk, L.stream()))
        ctx.save_for_backward(qkv, kv, gamma, att, A)
        ctx.meta = (dq, C_real, koff, voff, N, Nk)
        return y

    @staticmethod
    def backward(ctx, dy):
        qkv, kv, gamma, att, A = ctx.saved_tensors
        dq, C_real, koff, voff, N, Nk = ctx.meta
        dy = dy.contiguous()
        F_ = dy.shape[0]
        dqkv = torch.zeros_like(qkv)          # only the q columns are written here
        dkv = torch.zeros_like(kv)            # k and v columns
        dS = torch.empty_like(A)
        dgamma = torch.zeros(1, dtype=torch.float32, device=dy.device)
        L.check(L.lib().dvd_attention_kv_backward(L.dt(dy), L.ptr(qkv), qkv.shape[-1], dq, L.ptr(kv), kv.shape[-1], koff,
                                                  voff, L.ptr(dy), dy.shape[-1], C_real, L.ptr(gamma), L.ptr(att), L.ptr(A),
                                                  L.ptr(dS), L.ptr(dqkv), L.ptr(dkv), L.ptr(dgamma), C.c_longlong(F_), N, Nk,
                                                  L.stream()))
        return dy, dqkv, dkv, dgamma, None, None


class SeparableAttnCellFn(Function):
    """One SeparableAttnCell (Attention.py:61-111) on channels-last tensors: y = gamma * out + x, with q | k | v the
    columns [0,dq) | [koff,..) | [voff,..) of the fused projection `qkv` [B,T,W,H,ld]; axis 0 / 1 / 2 = T / W / H."""

    @staticmethod
    def forward(ctx, x, qkv, gamma, dq, C_real, axis):
        B, T, W, H, ldx = x.shape
        koff = K.pad8(dq)
        voff = 2 * koff
        N = T * W * H
        A = (T, W, H)[axis]
        dev = x.device
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        Qf, Kp, Vp = f32(B * dq * N), f32(B * dq * N // 2), f32(B * C_real * N // 2)
        ksel, vsel, att = u8(B * dq * N // 2), u8(B * C_real * N // 2), f32(B * A * (A // 2))
        y = torch.zeros_like(x) if ldx != C_real else torch.empty_like(x)
        L.check(L.lib().dvd_sepattn_forward(L.dt(x), L.ptr(qkv), qkv.shape[-1], dq, koff, voff, L.ptr(x), ldx, C_real,
                                            L.ptr(gamma), L.ptr(y), L.ptr(Qf), L.ptr(Kp), L.ptr(Vp), L.ptr(ksel), L.ptr(vsel),
                                            L.ptr(att), C.c_longlong(B), T, W, H, axis, L.stream()))
        ctx.save_for_backward(gamma, Qf, Kp, Vp, ksel, vsel, att)
        ctx.meta = (B, T, W, H, ldx, qkv.shape[-1], dq, C_real, koff, voff, axis, qkv.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        gamma, Qf, Kp, Vp, ksel, vsel, att = ctx.saved_tensors
        B, T, W, H, ldx, ldq, dq, C_real, koff, voff, axis, qdt = ctx.meta
        dy = dy.contiguous()
        dev = dy.device
        N = T * W * H
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        dO, dS = f32(B * C_real * N), torch.empty_like(att)
        dQf, dKp, dVp = torch.empty_like(Qf), torch.empty_like(Kp), torch.empty_like(Vp)
        dqkv = torch.zeros(B, T, W, H, ldq, dtype=qdt, device=dev)
        dgamma = torch.zeros(1, dtype=torch.float32, device=dev)
        L.check(L.lib().dvd_sepattn_backward(L.dt(dy), L.ptr(dy), ldx, C_real, dq, L.ptr(gamma), L.ptr(Qf), L.ptr(Kp), L.ptr(Vp),
                                             L.ptr(ksel), L.ptr(vsel), L.ptr(att), L.ptr(dO), L.ptr(dS), L.ptr(dQf), L.ptr(dKp),
                                             L.ptr(dVp), L.ptr(dqkv), ldq, koff, voff, L.ptr(dgamma), C.c_longlong(B), T, W, H,
                                             axis, L.stream()))
        return dy, dqkv, dgamma, None, None, None


class ProjectionHead(Function):
    """out[f] = b + sum_c h[f][c] * (w_lin[c]/s_l + embed[cls[f]][c]/s_e),  h = sum_p relu(feat)
    Discriminators.py:264-291 / 421-447 (both SN wrappers have their own sigma)."""

    @staticmethod
    def forward(ctx, feat, w_lin, b_lin, w_emb, cls32, sn_lin, sn_emb, C_real):
        F_ = feat.shape[0]
        P = feat.numel() // (F_ * feat.shape[-1])
        lib = L.lib()
        hsum = torch.empty(F_, C_real, dtype=torch.float32, device=feat.device)
        L.check(lib.dvd_relu_spatial_sum(L.dt(feat), L.ptr(feat), L.ptr(hsum), C.c_longlong(F_), P, C_real,
                                         feat.shape[-1], L.stream()))
        s_l = K.sn_power_iter(w_lin, *sn_lin)
        s_e = K.sn_power_iter(w_emb, *sn_emb)
        out = torch.empty(F_, dtype=torch.float32, device=feat.device)
        L.check(lib.dvd_proj_head_forward(L.ptr(hsum), L.ptr(w_lin), L.ptr(s_l), L.ptr(b_lin), L.ptr(w_emb), L.ptr(s_e),
                                          L.ptr(cls32), L.ptr(out), C.c_longlong(F_), C_real, L.stream()))
        ctx.save_for_backward(feat, hsum, w_lin, w_emb, cls32, s_l, s_e)
        ctx.sn = (sn_lin, sn_emb)
        ctx.params = (w_lin, b_lin, w_emb)
        ctx.meta = (F_, P, C_real)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, hsum, w_lin, w_emb, cls32, s_l, s_e = ctx.saved_tensors
        F_, P, C_real = ctx.meta
        sn_lin, sn_emb = ctx.sn
        lib = L.lib()
        dout = dout.contiguous()
        need_w = ctx.needs_input_grad[1]
        direct = need_w and _direct(*ctx.params)
        dh = torch.empty_like(hsum)
        g_lin = torch.zeros_like(w_lin) if need_w else None
        g_emb = torch.zeros_like(w_emb) if need_w else None
        g_b = (ctx.params[1].grad if direct else torch.zeros(1, dtype=torch.float32, device=feat.device)) if need_w else None
        L.check(lib.dvd_proj_head_backward(L.ptr(dout), L.ptr(hsum), L.ptr(w_lin), L.ptr(s_l), L.ptr(w_emb), L.ptr(s_e),
                                           L.ptr(cls32), L.ptr(dh), L.ptr(g_lin), L.ptr(g_emb), L.ptr(g_b),
                                           C.c_longlong(F_), C_real, L.stream()))
        dfeat = None
        if ctx.needs_input_grad[0]:
            dfeat = torch.empty_like(feat)
            L.check(lib.dvd_relu_spatial_sum_backward(L.dt(feat), L.ptr(dh), L.ptr(feat), L.ptr(dfeat), C.c_longlong(F_),
                                                      P, C_real, feat.shape[-1], L.stream()))
        dwl = dwe = None
        if direct:                 # spectral-norm backward adds straight into the persistent .grad buffers
            K.sn_backward(g_lin, w_lin, sn_lin[0], sn_lin[1], s_l, out=ctx.params[0].grad)
            K.sn_backward(g_emb, w_emb, sn_emb[0], sn_emb[1], s_e, out=ctx.params[2].grad)
            return dfeat, None, None, None, None, None, None, None
        if need_w:
            dwl = K.sn_backward(g_lin, w_lin, sn_lin[0], sn_lin[1], s_l)
            dwe = K.sn_backward(g_emb, w_emb, sn_emb[0], sn_emb[1], s_e)
        return dfeat, dwl, g_b, dwe, None, None, None, None


class AdvLoss(Function):
    """Trainer.calc_loss, trainer.py:114-121 -> scalar"""

    @staticmethod
    def forward(ctx, out, real_flag, hinge):
        out = out.contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=out.device)
        dout = torch.empty_like(out)
        L.check(L.lib().dvd_adv_loss(L.ptr(out), C.c_longlong(out.numel()), int(hinge), int(real_flag), L.ptr(loss),
                                     L.ptr(dout), C.c_float(1.0), L.stream()))
        ctx.save_for_backward(dout)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return dout * g, None, None


# ------------------------------------------------------------------ reference-layout helpers
class VidDownsample(Function):
    """utils.py:77-83"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return K.vid_downsample_raw(x.contiguous(), False, ctx.shape)

    @staticmethod
    def backward(ctx, g):
        return K.vid_downsample_raw(g.contiguous(), True, ctx.shape)


def _ids_to(ids, device):
    from .helpers import to_device_async
    return to_device_async(ids, device)


class GatherFrames(Function):
    """data[:, ids] for [B, T, ...] fp32 data (utils.py:63)"""

    @staticmethod
    def forward(ctx, x, ids):
        B, T = x.shape[:2]
        k = ids.numel()
        rows = (torch.arange(B, device=x.device).view(B, 1) * T + _ids_to(ids.view(1, k), x.device)).reshape(-1).int()
        Lr = x[0, 0].numel()
        ctx.save_for_backward(rows)
        ctx.shape = tuple(x.shape)
        out = K.row_copy(x.contiguous().view(B * T, Lr), rows, B * k, Lr, scatter=False)
        return out.view(B, k, *x.shape[2:])

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        B, T = ctx.shape[:2]
        Lr = g[0, 0].numel()
        out = K.row_copy(g.contiguous().view(-1, Lr), rows, B * T, Lr, scatter=True)
        return out.view(ctx.shape), None
