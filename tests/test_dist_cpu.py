"""world_size=2 gloo test (CPU) of the data-parallel pieces: batch sharding and the gradient exchange.
Each rank computes the discriminator gradients of ITS shard with the CPU oracle; after
GradExchange (all-reduce mean on one flat buffer) both ranks must hold the gradient of the GLOBAL batch
(loss means over equal shards average exactly; D has no batch coupling)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import load_golden, sub


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ds_grads(g, x, cls):
    from oracle import dvdgan_cpu as O
    sd = O.make_state(sub(sub(g, "ds"), "sd0"))
    loss = O.adv_loss(O.spatial_disc(sd, x, cls), True, "hinge")
    loss.backward()
    keys = sorted(k for k in sd if O.is_trainable(k))
    return keys, torch.cat([sd[k].grad.reshape(-1) for k in keys])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from dvd_gan_amd import dist as D
    r, w, dev = D.init_from_env("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    g = load_golden("f7_discriminators")
    x, cls = torch.as_tensor(g["ds.in.x"]), torch.as_tensor(g["ds.in.cls"])       # global batch of 2 clips
    xs, cs = D.shard(x, rank, world), D.shard(cls, rank, world)
    assert xs.shape[0] == 1
    _, flat = _ds_grads(g, xs, cs)
    ex = D.GradExchange()
    bucketed = flat.clone()
    ex.start("Ds", flat)
    ex.finish("Ds")
    # the generator's exchange goes in buckets (tail of the flat buffer first): same result as one all-reduce
    n = bucketed.numel()
    for lo, hi in ((2 * n // 3, n), (n // 3, 2 * n // 3), (0, n // 3)):
        ex.start_range("G", bucketed, lo, hi)
    ex.finish("G")
    assert torch.equal(bucketed, flat)
    # --- rank-0 broadcast of a model whose ranks were seeded differently (flat buffer + SN-style frozen parameter + buffers)
    torch.manual_seed(10 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    net[0].register_parameter("weight_u", torch.nn.Parameter(torch.randn(4), requires_grad=False))
    net[1].running_mean.normal_()
    flat_p = torch.cat([p.data.reshape(-1) for p in net.parameters() if p.requires_grad])
    off = 0
    for p in net.parameters():
        if p.requires_grad:
            p.data = flat_p[off:off + p.numel()].view(p.shape); off += p.numel()
    D.broadcast_state([net], [flat_p])
    state = torch.cat([v.reshape(-1).float() for v in net.state_dict().values()])
    both = [torch.empty_like(state) for _ in range(world)]
    torch.distributed.all_gather(both, state)
    assert torch.equal(both[0], both[1])
    seed = D.shared_seed()
    seeds = [None] * world
    torch.distributed.all_gather_object(seeds, seed)
    assert seeds[0] == seeds[1]
    # --- differentiable row gather: forward = concatenation in rank order, backward = sum over ranks of the row gradients
    xr = (torch.arange(6.).view(2, 3) + 10 * rank).requires_grad_(True)
    allr = D.AllGatherRows.apply(xr)
    assert torch.equal(allr.detach(), torch.cat([torch.arange(6.).view(2, 3), torch.arange(6.).view(2, 3) + 10]))
    (allr * (rank + 1) * torch.arange(1., 5.).view(4, 1)).sum().backward()
    want = (1 + 2) * torch.arange(1., 5.)[rank * 2:rank * 2 + 2].view(2, 1).expand(2, 3)
    assert torch.equal(xr.grad, want), (xr.grad, want)
    t = torch.ones(3, dtype=torch.float64) * (rank + 1)
    assert torch.equal(D.all_reduce_sum_(t), torch.full((3,), 3.0, dtype=torch.float64))
    out[rank] = flat.numpy().copy()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_exchange_equals_global_batch():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g = load_golden("f7_discriminators")
    _, full = _ds_grads(g, torch.as_tensor(g["ds.in.x"]), torch.as_tensor(g["ds.in.cls"]))
    np.testing.assert_allclose(out[0], out[1], rtol=0, atol=0)
    np.testing.assert_allclose(out[0], full.numpy(), rtol=2e-4, atol=2e-6)


def _worker4(rank, world, port, out):
    """World size 4: the generator's bucketed exchange in the Trainer's order (tail of the flat buffer first, uneven bucket
    sizes incl. an empty one), the row gather and the shared seed."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from dvd_gan_amd import dist as D
    r, w, dev = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n = 10007
    gen = torch.Generator().manual_seed(5)
    parts = torch.randn(world, n, generator=gen)                  # every rank knows every rank's gradient
    flat = parts[rank].clone()
    ex = D.GradExchange()
    hi = n
    for lo in (9000, 9000, 4097, 1, 0):                           # the on_ready hook's [lo, hi) ranges, an empty one included
        ex.start_range("G", flat, lo, hi)
        hi = min(hi, lo)
    ex.finish("G")
    assert torch.allclose(flat, parts.mean(0), rtol=0, atol=1e-6)
    rows = (torch.arange(3.).view(1, 3) + 10 * rank).requires_grad_(True)
    allr = D.AllGatherRows.apply(rows)
    assert torch.equal(allr.detach(), torch.cat([torch.arange(3.).view(1, 3) + 10 * q for q in range(world)]))
    (allr * (rank + 1)).sum().backward()
    assert torch.equal(rows.grad, torch.full((1, 3), float(sum(range(1, world + 1)))))
    seeds = [None] * world
    torch.distributed.all_gather_object(seeds, D.shared_seed())
    assert len(set(seeds)) == 1
    out[rank] = 1
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_four_rank_bucketed_exchange():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker4, args=(4, _free_port(), out), nprocs=4, join=True)
    assert sorted(out.keys()) == [0, 1, 2, 3]


def test_rccl_group_is_created_with_a_high_priority_stream():
    """RCCL's kernels run on ProcessGroupNCCL's own stream; it is high-priority only when the group is created with
    Options(is_high_priority_stream=True).  dist.init_from_env passes this object as pg_options for the nccl backend."""
    import inspect
    from dvd_gan_amd import dist as D
    opts = D.nccl_high_priority_options()
    if opts is None:
        import pytest
        pytest.skip("this torch build has no nccl backend")
    assert opts.is_high_priority_stream is True
    src = inspect.getsource(D.init_from_env)
    assert "pg_options" in src and "nccl_high_priority_options" in src
