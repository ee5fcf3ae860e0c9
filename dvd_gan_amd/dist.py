"""Data-parallel glue: one process per GPU, gradients averaged with RCCL all-reduce over xGMI.

The reference only has nn.DataParallel (trainer.py:353-359): single process, gradients summed
onto GPU 0, batch-norm statistics and SN power iterations per replica.  Here each rank owns
clips [r*B/N, (r+1)*B/N) of the global batch; after each of the three backward passes the flat
gradient buffer of that network (optim.FlatAdam.grad) is all-reduced on a side stream: D_s in one piece
(it overlaps the D_t forward/backward), D_t's overlaps the generator step's D_s forward, the generator's goes in
buckets that start as soon as the backward pass has left the corresponding ConvGRU stage (the buffer's tail is final
first).  SN u/v need no exchange because weights are identical on every rank (`broadcast_state` makes them so).

Two batch-norm semantics (Trainer(dp_mode=...)):
  "replica"  per-replica statistics and per-replica condition rows -- what nn.DataParallel does (default);
  "global"   cross-replica conditional batch norm (the reference's own TODO, Generator.py:57) plus the condition
             matrix gathered over the ranks, so that N ranks reproduce ONE process on the global batch exactly
             (including the reference's condition mis-ordering, which indexes the GLOBAL batch).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* when launched by
    torch.distributed.run; no-op for a single process.  Returns (rank, world_size, device)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL buffer sharing across processes)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        if os.environ.get("DVD_SHARE_GPU0"):          # test aid: several ranks on one GPU (gloo backend)
            local = 0
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local)
    backend = backend or os.environ.get("DVD_DIST_BACKEND")
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_cuda else "gloo")
        kw = {}
        if backend == "nccl" and exchange_priority():
            # RCCL's kernels run on ProcessGroupNCCL's OWN internal stream, whatever stream the collective was issued from
            # (the issuing stream is only fenced with events): that stream is high-priority only when the group is created
            # with this option.  OFF by default since round 6 (see exchange_priority).
            opts = nccl_high_priority_options()
            if opts is not None:
                kw["pg_options"] = opts
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, (torch.device("cuda", local) if use_cuda else torch.device("cpu"))


def nccl_high_priority_options():
    """ProcessGroupNCCL.Options(is_high_priority_stream=True), or None when this torch build has no nccl backend."""
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except (AttributeError, RuntimeError):
        return None


def exchange_priority():
    """DVD_EXCHANGE_PRIO=1: high-priority process-group stream + exchange stream.  Default 0 since round 6 -- the first
    measurement of the nccl branch (one-rank RCCL group, `bench.py --force-exchange`, B=64, two boxes): with the urgent
    level the step runs 523.9 / 536.5 ms against 453.4 / 477.8 plain (+15 / +12 %); with every stream of the data-parallel
    step at the normal level 458.5 / 478.9 ms (+1.1 / +0.2 %).  HIP has two levels here (priority_range (0, -1)): an urgent
    RCCL stream ranks above the step's own chain, and its event fences stall the chain far longer than the 0.8 ms the
    547 MB all-reduce kernel itself takes (tools: profiles/r06_forced_exchange.txt).  No multi-GPU hardware was available:
    with real peers the trade may differ, the switch stays."""
    return os.environ.get("DVD_EXCHANGE_PRIO", "0") == "1"


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def forced():
    """DVD_FORCE_EXCHANGE=1 (test / measurement switch): run the whole data-parallel machinery -- process group, rank-0
    broadcast, shared seed, gradient all-reduces on the exchange stream, the generator's bucket hooks and their side-stream
    fences, cross-replica sums in "global" mode -- even when the world is ONE rank.  A one-rank RCCL group executes every
    collective except the transport, so `tests/test_gpu_dist.py` can hold the nccl branch bit-equal to the plain step on a
    single GPU, and `bench.py --force-exchange` prices the exchange's host / fence overhead at B=64."""
    return os.environ.get("DVD_FORCE_EXCHANGE", "0") == "1"


def exchange_on():
    """True when gradients (and, in "global" mode, batch statistics / condition rows) go through torch.distributed."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def collective_device():
    """Device a tensor must live on to take part in a collective of the default group."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _avg_(t):
    """In-place mean over the ranks.  RCCL averages inside the collective (ReduceOp.AVG: no extra pass over the
    buffer); gloo has no AVG, so the CPU / test path divides afterwards."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(t)
        t.div_(dist.get_world_size())


def all_reduce_sum_(t):
    """In-place sum over the ranks on the current stream (cross-replica batch-norm sums: [2C] fp64 / fp32)."""
    dist.all_reduce(t)
    return t


class GradExchange:
    """Averages flat gradient buffers across ranks; asynchronous on a side stream when on GPU."""

    def __init__(self):
        self.world = world_size()
        self.active = exchange_on()           # world > 1, or a one-rank group with DVD_FORCE_EXCHANGE=1
        self.stream = None
        if self.active and torch.cuda.is_available():
            # The exchange is ISSUED from its own stream; that only orders the event fences around the collective.  The
            # all-reduce kernels themselves run on ProcessGroupNCCL's internal stream.  Both are normal-priority streams
            # unless DVD_EXCHANGE_PRIO=1 (exchange_priority: the urgent level measured 12-15 % slower on a one-rank group).
            hi = torch.cuda.Stream.priority_range()[1] if exchange_priority() else 0
            self.stream = torch.cuda.Stream(priority=hi)
        self.pending = {}

    def _launch(self, key, view, after=()):
        if self.stream is None:                       # CPU / gloo: synchronous
            _avg_(view)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        for s in after:                               # producers on other streams (the weight-gradient side stream): the EXCHANGE
            if s is not None:                         # waits for them, the step's chain does not
                self.stream.wait_stream(s)
        with torch.cuda.stream(self.stream):
            _avg_(view)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending.setdefault(key, []).append(ev)

    def start(self, key, flat_grad, after=()):
        """Begin averaging `flat_grad` (in place).  Call finish(key) before the buffer is read.  `after`: further streams
        whose queued work produces the buffer (the exchange stream waits for them; the caller's stream is not fenced)."""
        if self.active:
            self._launch(key, flat_grad, after)

    def start_range(self, key, flat_grad, lo, hi, after=()):
        """Begin averaging flat_grad[lo:hi] (a bucket); several ranges may be pending under one key.  Every rank must
        issue the same ranges in the same order."""
        if self.active and hi > lo:
            self._launch(key, flat_grad[lo:hi], after)

    def finish(self, key):
        for e in self.pending.pop(key, ()):
            torch.cuda.current_stream().wait_event(e)


def broadcast_state(nets, flats=()):
    """Make every rank start from rank 0's model: the flat parameter buffers (`flats`), every parameter that is not
    a view of one of them (SN u / v are requires_grad=False parameters) and all buffers (batch-norm statistics).
    Without this, ranks seeded differently (needed for distinct z per rank) would average gradients of DIFFERENT
    models and never agree -- the reference's nn.DataParallel replicates from GPU 0 every forward instead."""
    if not exchange_on():
        return
    with torch.no_grad():
        for f in flats:
            dist.broadcast(f, 0)
        lo_hi = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in flats]
        for net in nets:
            for t in list(net.parameters()) + list(net.buffers()):
                if any(lo <= t.data_ptr() < hi for lo, hi in lo_hi):
                    continue
                if t.dim() == 0:                      # gloo cannot broadcast 0-d integer tensors in place on every build
                    v = t.reshape(1).clone()
                    dist.broadcast(v, 0)
                    t.copy_(v[0])
                else:
                    dist.broadcast(t, 0)


def shared_seed():
    """One random 63-bit seed agreed by all ranks (drawn on rank 0): seeds the frame-id generator, so every rank samples
    the same k frame ids per step while z / labels stay per rank."""
    seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if exchange_on():
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        s = seed.to(dev)
        dist.broadcast(s, 0)
        seed = s.cpu()
    return int(seed)


class AllGatherRows(torch.autograd.Function):
    """[b, ...] per rank -> [world * b, ...] (rank-major), differentiable: the gradient of a rank's rows is the SUM over
    the ranks of the gradients they hold for those rows."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out = x.new_empty((dist.get_world_size() * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x) if dist.get_backend() == "nccl" else \
            dist.all_gather(list(out.chunk(dist.get_world_size())), x)
        ctx.b = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g)
        r = dist.get_rank()
        return g[r * ctx.b:(r + 1) * ctx.b]


def shard(t, rank, world):
    """Rows [rank*B/N, (rank+1)*B/N) of a global batch."""
    per = t.shape[0] // world
    return t[rank * per:(rank + 1) * per]
