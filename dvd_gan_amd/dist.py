"""Data-parallel glue: one process per GPU, gradients averaged with RCCL all-reduce over xGMI.

The reference only has nn.DataParallel (trainer.py:353-359): single process, gradients summed
onto GPU 0, batch-norm statistics and SN power iterations per replica.  Here each rank owns
clips [r*B/N, (r+1)*B/N) of the global batch; after each of the three backward passes the flat
gradient buffer of that network (optim.FlatAdam.grad) is all-reduced on a side stream: D_s in one piece
(it overlaps the D_t forward/backward), the generator's in buckets that start as soon as the backward
pass has left the corresponding ConvGRU stage (the buffer's tail is final first).  Batch-norm statistics stay per replica (= DataParallel
semantics); SN u/v need no exchange because weights are identical on every rank.
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
        torch.cuda.set_device(local)
    backend = backend or os.environ.get("DVD_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    return rank, world, (torch.device("cuda", local) if use_cuda else torch.device("cpu"))


class GradExchange:
    """Averages flat gradient buffers across ranks; asynchronous on a side stream when on GPU."""

    def __init__(self):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.stream = torch.cuda.Stream() if (self.world > 1 and torch.cuda.is_available()) else None
        self.pending = {}

    def start(self, key, flat_grad):
        """Begin averaging `flat_grad` (in place).  Call finish(key) before the buffer is read."""
        if self.world == 1:
            return
        if self.stream is None:                       # CPU / gloo: synchronous
            dist.all_reduce(flat_grad)
            flat_grad.div_(self.world)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(flat_grad)
            flat_grad.div_(self.world)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending[key] = ev

    def start_range(self, key, flat_grad, lo, hi):
        """Begin averaging flat_grad[lo:hi] (a bucket); several ranges may be pending under one key.  Every rank must
        issue the same ranges in the same order."""
        if self.world == 1 or hi <= lo:
            return
        view = flat_grad[lo:hi]
        if self.stream is None:
            dist.all_reduce(view)
            view.div_(self.world)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(view)
            view.div_(self.world)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending.setdefault(key, [])
        if not isinstance(self.pending[key], list):
            self.pending[key] = [self.pending[key]]
        self.pending[key].append(ev)

    def finish(self, key):
        ev = self.pending.pop(key, None)
        if ev is None:
            return
        for e in (ev if isinstance(ev, list) else [ev]):
            torch.cuda.current_stream().wait_event(e)


def shard(t, rank, world):
    """Rows [rank*B/N, (rank+1)*B/N) of a global batch."""
    per = t.shape[0] // world
    return t[rank * per:(rank + 1) * per]
