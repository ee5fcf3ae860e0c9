"""Data-parallel Trainer on the device: N ranks drive the real HIP Trainer through one step.

  * two ranks sharing ONE GPU over gloo (CUDA tensors): dist.init_from_env, rank-0 broadcast of the initial model, batch
    sharding and GradExchange's side-stream ordering -- ranks stay in lock step, the discriminators (no batch coupling)
    match a single-process step on the global batch;
  * dp_mode="global" (cross-replica conditional batch norm + gathered condition rows): the two ranks reproduce ONE
    process on the global batch -- six losses and every gradient of G, D_s and D_t;
  * the same over RCCL (backend nccl, one rank per GPU) whenever at least two GPUs are visible;
  * a ONE-rank RCCL group with the exchange forced on (DVD_FORCE_EXCHANGE=1, dist.forced): the nccl branch of dist.py --
    ReduceOp.AVG, the high-priority process-group options, all_gather_into_tensor, the bucket hooks and the stream fences,
    everything except the transport -- must leave a bf16 step at the benchmark's widths BIT-EQUAL to the plain step.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import argparse, os, sys, torch
sys.path.insert(0, sys.argv[1])
from dvd_gan_amd import dist as D
from dvd_gan_amd.train_step import Trainer
mode = sys.argv[3]
rank, world, dev = D.init_from_env()
cfg = argparse.Namespace(adv_loss="hinge", z_dim=16, g_chn=2, ds_chn=2, dt_chn=2, n_frames=8, lr_schr="const",
                         total_epoch=1, d_iters=1, batch_size=4 // world, g_lr=2e-3, d_lr=2e-3, beta1=0.0, beta2=0.9,
                         n_class=3, k_sample=4)
torch.manual_seed(100 * rank)          # DIFFERENT initial weights per rank: the Trainer must broadcast rank 0's
tr = Trainer([], cfg, device=dev, compute_dtype=torch.float32, dp_mode=mode)
if world == 1:                         # single process: the model rank 0 of the 2-rank run starts from
    pass
g = torch.Generator().manual_seed(1)
real = torch.rand(4, 3, 8, 64, 64, generator=g) * 2 - 1
labels = torch.randint(0, 3, (4,), generator=g)
z = torch.randn(4, 16, generator=g); zc = torch.randint(0, 3, (4,), generator=g)
draws = {"perm_real": torch.arange(8), "z": D.shard(z, rank, world), "z_class": D.shard(zc, rank, world),
         "perm_fake": torch.arange(8).flip(0)}
grads = {}
for tag, net, opt in (("Ds", tr.D_s, tr.ds_optimizer), ("Dt", tr.D_t, tr.dt_optimizer), ("G", tr.G, tr.g_optimizer)):
    def stepper(net=net, tag=tag, orig=opt.step):
        grads[tag] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}
        orig()
    opt.step = stepper
ids = [tr.frame_gen.initial_seed()] if tr.frame_gen is not None else []
losses = tr.train_step(D.shard(real, rank, world), D.shard(labels, rank, world), draws)
torch.cuda.synchronize()
out = {"Ds": {k: v.detach().cpu() for k, v in tr.D_s.state_dict().items()},
       "Dt": {k: v.detach().cpu() for k, v in tr.D_t.state_dict().items()},
       "G": {k: v.detach().cpu() for k, v in tr.G.state_dict().items()},
       "losses": [float(v.detach()) for v in losses], "grads": grads, "frame_seed": ids}
torch.save(out, sys.argv[2] + f".{mode}.{world}.{rank}")
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


def _run(world, tmp, mode="replica", backend="gloo"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = os.path.join(tmp, "worker.py")
    open(script, "w").write(WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if backend == "gloo":
            env.update(DVD_SHARE_GPU0="1", DVD_DIST_BACKEND="gloo")
        log = open(os.path.join(tmp, f"worker.{mode}.{world}.{r}.log"), "w")
        procs.append(subprocess.Popen([sys.executable, script, ROOT, os.path.join(tmp, "out"), mode], env=env, stdout=log, stderr=log))
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    except BaseException:
        for p in procs:                                    # a stuck rank must not outlive the test (and hold the GPU / the port)
            if p.poll() is None:
                p.kill()
        for r in range(world):
            print(f"---- rank {r} of {world} ({mode}, {backend})")
            print(open(os.path.join(tmp, f"worker.{mode}.{world}.{r}.log")).read()[-3000:])
        raise
    return [torch.load(os.path.join(tmp, f"out.{mode}.{world}.{r}")) for r in range(world)]


def _single(tmp):
    """One process on the global batch, started from the weights rank 0 draws (seed 0)."""
    return _run(1, tmp)[0]


def _check_lock_step(two):
    for net in ("G", "Ds", "Dt"):
        for k, v in two[0][net].items():
            if "running_" in k:
                continue                  # batch-norm statistics are per replica (= nn.DataParallel semantics)
            assert torch.equal(v, two[1][net][k]), (net, k)            # ranks stay in lock step
    assert two[0]["frame_seed"] == two[1]["frame_seed"] and len(two[0]["frame_seed"]) == 1


def _check_replica_mode(two, one):
    _check_lock_step(two)
    # D_s: its update only depends on D_s gradients (means over equal shards) when the fake inputs agree;
    # fake videos differ between 1 and 2 ranks only through per-replica batch-norm statistics in G, so the
    # REAL-data half is exact; compare D parameters loosely and the real-data losses tightly.
    assert abs(0.5 * (two[0]["losses"][0] + two[1]["losses"][0]) - one["losses"][0]) < 1e-5     # ds_real
    assert abs(0.5 * (two[0]["losses"][2] + two[1]["losses"][2]) - one["losses"][2]) < 1e-5     # dt_real
    for net in ("Ds", "Dt"):
        for k, v in one[net].items():
            if v.is_floating_point() and not k.endswith(("weight_u", "weight_v")):
                assert float((two[0][net][k] - v).abs().max()) < 5e-3, (net, k)      # <= 2.5 Adam steps of lr


def _check_global_mode(two, one):
    """N ranks == 1 rank: losses (mean of the per-rank means), every gradient after the exchange, BN running statistics."""
    n = len(two)
    for r in range(1, n):
        for k, v in two[0]["G"].items():
            assert torch.equal(v, two[r]["G"][k]), k                      # now the BN buffers agree as well
    for i in range(6):
        assert abs(sum(t["losses"][i] for t in two) / n - one["losses"][i]) < 2e-5, i
    worst = 0.0
    for net in ("Ds", "Dt", "G"):
        ref = one["grads"][net]
        scale = max(float(v.abs().max()) for v in ref.values())
        for k, v in ref.items():
            got = two[0]["grads"][net][k]
            for r in range(1, n):
                assert torch.equal(got, two[r]["grads"][net][k]), (net, k)
            if float(v.abs().max()) < 1e-4 * scale:
                continue
            r = float((got.double() - v.double()).norm() / v.double().norm())
            worst = max(worst, r)
            assert r < 2e-3, (net, k, r)
    for k, v in one["G"].items():
        if "running_" in k:
            assert float((two[0]["G"][k] - v).abs().max()) < 1e-5, k
    return worst


def test_two_ranks_one_gpu_match_each_other_and_the_global_batch(tmp_path):
    _check_replica_mode(_run(2, str(tmp_path)), _single(str(tmp_path)))


def test_two_ranks_global_mode_equal_one_process_on_the_global_batch(tmp_path):
    _check_global_mode(_run(2, str(tmp_path), mode="global"), _single(str(tmp_path)))


def test_four_ranks_global_mode_equal_one_process_on_the_global_batch(tmp_path):
    """World size 4 (one clip per rank, four ranks on one GPU over gloo): the generator's gradient goes in its four buckets
    (stage-boundary hooks), cross-replica batch norm sums over four replicas, the condition rows of four ranks are
    gathered -- and the result is still ONE process on the global batch of four clips."""
    _check_global_mode(_run(4, str(tmp_path), mode="global"), _single(str(tmp_path)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL path needs two GPUs (one rank per GPU)")
@pytest.mark.parametrize("mode", ["replica", "global"])
def test_two_gpus_over_rccl(tmp_path, mode):
    two = _run(2, str(tmp_path), mode=mode, backend="nccl")
    one = _single(str(tmp_path))
    if mode == "replica":
        _check_replica_mode(two, one)
    else:
        _check_global_mode(two, one)


FORCED_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import importlib.util
from dvd_gan_amd import dist as D
mode, forced = sys.argv[3], sys.argv[4] == "1"
if forced:
    rank, world, dev = D.init_from_env("nccl")
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and world == 1
spec = importlib.util.spec_from_file_location("repro_probe", os.path.join(sys.argv[1], "tools", "repro_probe.py"))
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
from dvd_gan_amd import train_step
if mode == "global":                      # (tools/repro_probe.build constructs Trainer(...) with the default dp_mode)
    orig = train_step.Trainer.__init__
    def init(self, *a, **kw):
        kw["dp_mode"] = "global"
        orig(self, *a, **kw)
    train_step.Trainer.__init__ = init
    probe.Trainer = train_step.Trainer
calls = {"n": 0}
if forced:
    real_ar = torch.distributed.all_reduce
    def counting(*a, **kw):
        calls["n"] += 1
        return real_ar(*a, **kw)
    torch.distributed.all_reduce = counting
losses, state = probe.run(32, 8, 2, 7, 2)
torch.cuda.synchronize()
torch.save({"losses": losses, "state": {k: v.cpu() for k, v in state.items()}, "all_reduces": calls["n"]}, sys.argv[2])
if forced:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


def _forced_run(tmp, mode, forced):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = os.path.join(tmp, "forced_worker.py")
    open(script, "w").write(FORCED_WORKER)
    out = os.path.join(tmp, f"forced.{mode}.{int(forced)}")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", DVD_FORCE_EXCHANGE="1" if forced else "0")
    env.pop("DVD_DIST_BACKEND", None)
    assert subprocess.call([sys.executable, script, ROOT, out, mode, "1" if forced else "0"], env=env, timeout=900) == 0
    return torch.load(out)


@pytest.mark.parametrize("mode", ["replica", "global"])
def test_one_rank_rccl_group_with_forced_exchange_equals_plain_step_bitwise(tmp_path, mode):
    """trainer.py:353-359's replacement over the nccl backend, on ONE GPU: two bf16 steps at ch=32 (the reproducibility test's
    shape) with every gradient all-reduced by RCCL (AVG over one rank), the generator's gradient in its four buckets behind the
    stage hooks, and in "global" mode the batch-norm sums / condition rows through all_reduce / all_gather_into_tensor.  Losses,
    every gradient the optimizers saw and every parameter / buffer afterwards equal the plain single-process step bit for bit."""
    plain = _forced_run(str(tmp_path), "replica", False)
    forced = _forced_run(str(tmp_path), mode, True)
    # per step: D_s + D_t + four generator buckets (+ the cross-replica statistics in global mode)
    assert forced["all_reduces"] >= 2 * 6, forced["all_reduces"]
    if mode == "global":
        assert forced["all_reduces"] > 2 * 6 + 16
    assert plain["losses"] == forced["losses"], (plain["losses"], forced["losses"])
    assert len(plain["state"]) > 600 and plain["state"].keys() == forced["state"].keys()
    bad = [k for k in plain["state"] if not torch.equal(plain["state"][k], forced["state"][k])]
    assert not bad, bad[:10]
