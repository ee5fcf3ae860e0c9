"""The G / D_s / D_t training step of trainer.py:213-307 on the HIP path.

`Trainer(data_loader, config)` keeps the reference's constructor, `build_model`, `select_opt_schr`,
`calc_loss`, `reset_grad`, `train` names and the exact step order:
    perm(real) -> z -> z_class -> perm(fake)            (CPU default generator, trainer.py:233-242)
    D_s(real) , D_s(fake.detach)  -> backward -> ds Adam
    D_t(real↓), D_t(fake↓.detach) -> backward -> dt Adam
    D_s(fake), D_t(fake↓) on the UPDATED discriminators -> relu(1 - out) -> backward -> g Adam
Differences that do not change results: the dead D weight-gradients of the generator step are not
computed; Adam is one fused launch per network; gradients are exchanged with RCCL (dist.py).
Out of scope: tensorboard logging, sample grids, dataset loaders.
"""
import os
import time

import torch

from . import functional as Fn
from . import lib as L
from . import dist as D
from .dist import GradExchange
from .disc_nets import SpatialDiscriminator, TemporalDiscriminator
from .gen_net import Generator
from .helpers import denorm, draw_frame_ids, sample_k_frames, to_device_async, vid_downsample
from .optim import FlatAdam


class _StepLR:
    """lr schedule arithmetic of trainer.py:142-176 ('const' | 'step' | 'exp' | 'multi')."""

    def __init__(self, opt, kind, base_lr):
        self.opt, self.kind, self.base, self.n = opt, kind, base_lr, 0

    def step(self, metric=None):
        self.n += 1
        n, lr = self.n, self.base
        if self.kind == "step":
            lr = self.base * 0.98 ** (n // 500)
        elif self.kind == "exp":
            lr = self.base * 0.9999 ** n
        elif self.kind == "multi":
            lr = self.base * 0.3 ** ((n >= 10000) + (n >= 30000))
        self.opt.param_groups[0]["lr"] = lr

    def get_lr(self):
        return [self.opt.param_groups[0]["lr"]]


class _PlateauLR:
    """lr_schr='reduce' (trainer.py:158-176): ReduceLROnPlateau(mode='min', factor=lr_decay, patience=100,
    threshold=1e-4 'rel', cooldown=0, min_lr=1e-10, eps=1e-8) -- same arithmetic as torch's scheduler.  The reference
    calls `.step()` WITHOUT the metric (trainer.py:254,270,308), which raises TypeError at its first iteration, so that
    mode never ran there; here each network's scheduler is fed that network's loss of the step (what the configuration
    evidently intends).  Reading the loss costs one host sync per optimizer phase -- only in this mode."""

    def __init__(self, opt, factor, base_lr, patience=100, threshold=1e-4, min_lr=1e-10, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.opt, self.factor, self.patience, self.threshold, self.min_lr, self.eps = opt, factor, patience, threshold, min_lr, eps
        self.best, self.bad = float("inf"), 0
        self.opt.param_groups[0]["lr"] = base_lr

    def step(self, metric):
        if metric is None:
            raise TypeError("step() missing 1 required positional argument: 'metrics'")
        if D.exchange_on():
            # data parallel: every rank must take the SAME lr decision or the replicas' parameters drift apart for good
            # (only gradients are exchanged) -- the schedulers see the mean of the ranks' losses.  (A Python float is accepted
            # like in the single-process path; every rank must call step() every iteration: it is a collective.)
            dev = metric.device if isinstance(metric, torch.Tensor) else D.collective_device()
            metric = torch.as_tensor(metric, dtype=torch.float32).detach().to(dev).clone().reshape(1)
            torch.distributed.all_reduce(metric, op=torch.distributed.ReduceOp.SUM)
            metric = metric / D.world_size()
        m = float(metric)
        if m < self.best * (1.0 - self.threshold):
            self.best, self.bad = m, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            old = self.opt.param_groups[0]["lr"]
            new = max(old * self.factor, self.min_lr)
            if old - new > self.eps:
                self.opt.param_groups[0]["lr"] = new
            self.bad = 0

    def get_lr(self):
        return [self.opt.param_groups[0]["lr"]]


class Trainer(object):
    def __init__(self, data_loader, config, device=None, compute_dtype=torch.bfloat16, latent_dim=4, dp_mode="replica"):
        """dp_mode (data-parallel runs only): "replica" = per-replica batch-norm statistics and condition rows, the
        semantics of the reference's nn.DataParallel (trainer.py:353-359); "global" = cross-replica conditional batch
        norm + gathered conditions: N ranks reproduce one process on the global batch."""
        if dp_mode not in ("replica", "global"):
            raise ValueError("dp_mode must be 'replica' or 'global'")
        self.dp_mode = dp_mode
        self.data_loader = data_loader
        c = self.config = config
        self.adv_loss, self.z_dim = c.adv_loss, c.z_dim
        self.g_chn, self.ds_chn, self.dt_chn = c.g_chn, c.ds_chn, c.dt_chn
        self.n_frames, self.lr_schr = c.n_frames, c.lr_schr
        self.total_epoch, self.d_iters, self.batch_size = c.total_epoch, c.d_iters, c.batch_size
        self.g_lr, self.d_lr, self.beta1, self.beta2 = c.g_lr, c.d_lr, c.beta1, c.beta2
        self.n_class, self.k_sample = c.n_class, c.k_sample
        self.lr_decay = getattr(c, "lr_decay", 0.9999)
        self.pretrained_model = getattr(c, "pretrained_model", None)
        self.model_save_path = os.path.join(getattr(c, "model_save_path", "./models"), getattr(c, "version", ""))
        self.model_save_epoch = getattr(c, "model_save_epoch", 0)
        self.log_epoch = getattr(c, "log_epoch", 1)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.compute_dtype, self.latent_dim = compute_dtype, latent_dim
        Fn.direct_weight_grads(True)          # weight gradients accumulate into FlatAdam's buffers on a side stream
        self.exchange = GradExchange()
        # The step's dependent chain runs on a high-priority stream (ahead of the bulk weight-gradient stream) -- in a
        # single-process run.  In a data-parallel run the gradient exchange takes the urgent level instead and the chain
        # stays on the caller's stream, so RCCL's kernels are not queued behind every launch of the step while an exchange
        # is pending.  DVD_CHAIN_PRIO=1 / 0 forces the high-priority chain on / off.
        prio = os.environ.get("DVD_CHAIN_PRIO", "auto")
        self._chain = None
        self._aux = None          # stream of the discriminator passes over the real clips (see _train_step)
        if torch.cuda.is_available() and (prio == "1" or (prio == "auto" and not self.exchange.active)):
            self._chain = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
            # opt-in: -1.6 ms of 537 at 64 x 64, but the step DOUBLES at 48 x 128 x 128 (2002 -> 4043 ms; 175 GB of activations: blocks
            # freed on the second stream are not reusable by the first and the allocator falls back to synchronising frees)
            if not self.exchange.active and os.environ.get("DVD_D_REAL_EARLY", "0") == "1":
                self._aux = torch.cuda.Stream()
        self.rank = torch.distributed.get_rank() if self.exchange.active else 0
        self.build_model()
        if self.pretrained_model:
            self.load_pretrained_model()
        # data parallel: every rank continues from rank 0's model (parameters, SN u / v, BN statistics); frame ids come
        # from a generator all ranks seed alike, z / labels from each rank's own default generator
        self._sync_replicas()
        self.frame_gen = torch.Generator().manual_seed(D.shared_seed()) if self.exchange.active else None
        # ... and z / z_class of rank r from a generator of its own: distinct noise per replica even when every rank was
        # seeded alike (equal seeds on all ranks would train every replica on the same draws).  Its seed mixes the rank with
        # config.seed when the configuration has one, else with a value DRAWN from the caller's default generator -- so
        # torch.manual_seed(...) before constructing the Trainer steers it, two runs with different seeds draw different
        # noise, and a resumed run (which re-seeds or not as the caller likes) does not replay a fixed stream.
        # A single process keeps the default generator -- the reference's behaviour (trainer.py:84-88, 236-240).
        self.noise_gen = None
        if self.exchange.active:
            base = getattr(c, "seed", None)
            if base is None:
                base = int(torch.randint(0, 2 ** 31 - 1, (1,)))
            self.noise_gen = torch.Generator().manual_seed((int(base) * 1000003 + 7919 * self.rank + 1) % (2 ** 63 - 1))

    # ---- trainer.py:345-366
    def build_model(self):
        dt = self.compute_dtype
        c = self.config
        self.G = Generator(self.z_dim, self.latent_dim, self.n_class, self.g_chn, self.n_frames, compute_dtype=dt,
                           self_attn=getattr(c, "g_self_attn", False), sep_attn=getattr(c, "g_sep_attn", False)).to(self.device)
        self.D_s = SpatialDiscriminator(self.ds_chn, self.n_class, compute_dtype=dt).to(self.device)
        self.D_t = TemporalDiscriminator(self.dt_chn, self.n_class, compute_dtype=dt).to(self.device)
        if self.exchange.active and self.dp_mode == "global":
            from .sn_layers import ConditionalNorm
            self.G.dp_global = True
            for m in self.G.modules():
                if isinstance(m, ConditionalNorm):
                    m.replicas = (self.exchange.world, D.all_reduce_sum_)
        self.select_opt_schr()

    def _sync_replicas(self):
        D.broadcast_state((self.G, self.D_s, self.D_t),
                          (self.g_optimizer.flat, self.ds_optimizer.flat, self.dt_optimizer.flat))

    # ---- trainer.py:134-176
    def select_opt_schr(self):
        betas = (self.beta1, self.beta2)
        self.g_optimizer = FlatAdam(self.G.parameters(), self.g_lr, betas)
        # (the optional attention blocks sit at the END of the parameter order but finish their gradients late in the
        #  backward pass: with them the generator's gradient goes in one piece)
        self.G.dp_hooks = self.exchange.active and os.environ.get("DVD_DP_HOOKS", "1") != "0" and not (hasattr(self.G, "self_attn") or hasattr(self.G, "sep_attn"))
        # offset of the first trainable parameter of generator module conv.k in the flat buffers (gradient buckets)
        self._g_bounds, off = {}, 0
        for name, prm in self.G.named_parameters():
            if not prm.requires_grad:
                continue
            if name.startswith("conv."):
                self._g_bounds.setdefault(int(name.split(".")[1]), off)
            off += prm.numel()
        self.ds_optimizer = FlatAdam(self.D_s.parameters(), self.d_lr, betas)
        self.dt_optimizer = FlatAdam(self.D_t.parameters(), self.d_lr, betas)
        if self.lr_schr in ("const", "step", "exp", "multi"):
            self.g_lr_scher = _StepLR(self.g_optimizer, self.lr_schr, self.g_lr)
            self.ds_lr_scher = _StepLR(self.ds_optimizer, self.lr_schr, self.d_lr)
            self.dt_lr_scher = _StepLR(self.dt_optimizer, self.lr_schr, self.d_lr)
        else:                                   # trainer.py:158-176: anything else selects ReduceLROnPlateau
            self.g_lr_scher = _PlateauLR(self.g_optimizer, self.lr_decay, self.g_lr)
            self.ds_lr_scher = _PlateauLR(self.ds_optimizer, self.lr_decay, self.d_lr)
            self.dt_lr_scher = _PlateauLR(self.dt_optimizer, self.lr_decay, self.d_lr)
        self._plateau = not isinstance(self.g_lr_scher, _StepLR)

    # ---- trainer.py:114-121
    def calc_loss(self, x, real_flag):
        return Fn.AdvLoss.apply(x, bool(real_flag), self.adv_loss == "hinge")

    # ---- trainer.py:84-88
    def label_sample(self):
        return to_device_async(torch.randint(low=0, high=self.n_class, size=(self.batch_size,), generator=self.noise_gen), self.device)

    # ---- trainer.py:384-387
    def reset_grad(self):
        self.ds_optimizer.zero_grad()
        self.dt_optimizer.zero_grad()
        self.g_optimizer.zero_grad()

    def _freeze_d(self, flag):
        self.D_s._set_train_weights(not flag)
        self.D_t._set_train_weights(not flag)

    # ---- trainer.py:223-307, one iteration (d_iters D updates + one G update)
    def _check_labels(self, labels):
        """Class ids index embedding tables on the device (no bounds check there): reject a label outside
        [0, n_class) while it is still on the host, like the IndexError nn.Embedding raises in the reference."""
        if not labels.numel():
            return labels
        if labels.is_cuda and getattr(self, "_labels_registered", None) is labels and self._labels_version == labels._version:
            return labels           # a buffer the caller registered as persistent and has not written since (bench.py)
        # (a device tensor costs a host sync to inspect -- labels normally arrive on the host (DataLoader, label_sample)
        #  and take the free path; nothing is cached by address: a fresh tensor can reuse a freed block)
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= self.n_class:
            raise IndexError(f"class id out of range for n_class={self.n_class}: [{lo}, {hi}]")
        return labels

    def register_label_buffer(self, labels):
        """Declare a DEVICE label tensor the caller re-uses unchanged every step: it is validated now and not again while
        its version counter stands (saves the host sync of the range check in a steady-state loop)."""
        self._labels_registered = None
        self._check_labels(labels)
        self._labels_registered, self._labels_version = labels, labels._version

    def train_step(self, real_videos, real_labels, draws=None, hidden=None):
        """real_videos [B,3,T,H,W], real_labels [B].  `draws` (tests): dict with the reference's RNG
        draws perm_real / z / z_class / perm_fake.  `hidden`: initial ConvGRU states for the generator
        (Generator.forward, frame-conditional variant).  Returns the six loss terms as device scalars:
        ds_real, ds_fake, dt_real, dt_fake, g_s, g_t.
        The step's dependent chain runs on a HIGH-priority stream so its (often small) launches are dispatched ahead of
        the bulk weight-gradient work queued on the normal-priority side stream; the caller's stream is ordered before
        and after the step, so nothing changes for it."""
        if self._chain is None:
            return self._train_step(real_videos, real_labels, draws, hidden)
        outer = torch.cuda.current_stream()
        self._chain.wait_stream(outer)
        with torch.cuda.stream(self._chain):
            out = self._train_step(real_videos, real_labels, draws, hidden)
        outer.wait_stream(self._chain)
        for v in out:
            v.record_stream(outer)
        return out

    def _train_step(self, real_videos, real_labels, draws=None, hidden=None):
        L.reset_gru_tickets(current_stream_only=True)         # split-K tickets start every step from zero (lib.reset_gru_tickets)
        real_videos = to_device_async(real_videos, self.device).permute(0, 2, 1, 3, 4).contiguous()
        real_labels = to_device_async(self._check_labels(real_labels), self.device)
        T, k = self.n_frames, self.k_sample
        ex = self.exchange
        fg = self.frame_gen
        draws_all = draws
        for _ in range(self.d_iters):
            if isinstance(draws_all, (list, tuple)):              # test aid: one dict of draws per discriminator iteration
                draws = draws_all[_]
            ids_real = draw_frame_ids(T, k, fg) if draws is None else torch.as_tensor(draws["perm_real"])[:k].sort()[0]
            real_s = sample_k_frames(real_videos, T, k, ids_real)
            z = to_device_async(torch.randn(self.batch_size, self.z_dim, generator=self.noise_gen) if draws is None
                                else torch.as_tensor(draws["z"]), self.device)
            z_class = self.label_sample() if draws is None else to_device_async(self._check_labels(torch.as_tensor(draws["z_class"])), self.device)
            ex.finish("G")
            early = self._aux is not None
            if early:
                # The discriminator passes over the REAL clips need nothing from the generator: they run on a second stream beside
                # the 4 x 4 / 8 x 8 time loops at the start of the generator forward, the one phase of the step with idle CUs and
                # no weight-gradient work to fill them.  Same arithmetic in the same per-network order (real before fake, so the
                # spectral-norm state advances as in the reference); autograd runs their backward nodes on that stream too.
                cur = torch.cuda.current_stream()
                self._aux.wait_stream(cur)
                with torch.cuda.stream(self._aux):
                    ds_loss_real = self.calc_loss(self.D_s(real_s, real_labels), True)
                    real_d = vid_downsample(real_videos)
                    dt_loss_real = self.calc_loss(self.D_t(real_d, real_labels), True)
                for t_ in (real_s, real_videos, real_labels):
                    t_.record_stream(self._aux)
            fake_videos = self.G(z, z_class, hidden)
            ids_fake = draw_frame_ids(T, k, fg) if draws is None else torch.as_tensor(draws["perm_fake"])[:k].sort()[0]
            fake_s = sample_k_frames(fake_videos, T, k, ids_fake)
            # ---------------- D_s
            if early:
                cur.wait_stream(self._aux)
                for t_ in (ds_loss_real, dt_loss_real, real_d):
                    t_.record_stream(cur)
            else:
                ds_loss_real = self.calc_loss(self.D_s(real_s, real_labels), True)
            ds_loss_fake = self.calc_loss(self.D_s(fake_s.detach(), z_class), False)
            self.reset_grad()
            (ds_loss_real + ds_loss_fake).backward()
            Fn.join_side()
            ex.start("Ds", self.ds_optimizer.grad)
            # ---------------- D_t (its forward/backward overlaps the D_s gradient exchange)
            fake_d = vid_downsample(fake_videos)
            if not early:
                real_d = vid_downsample(real_videos)
                dt_loss_real = self.calc_loss(self.D_t(real_d, real_labels), True)
            dt_loss_fake = self.calc_loss(self.D_t(fake_d.detach(), z_class), False)
            ex.finish("Ds")
            self.ds_optimizer.step()
            self.ds_lr_scher.step((ds_loss_real + ds_loss_fake) if self._plateau else None)
            self.dt_optimizer.zero_grad()
            (dt_loss_real + dt_loss_fake).backward()
            Fn.join_side()
            ex.start("Dt", self.dt_optimizer.grad)
            last = _ == self.d_iters - 1
            if not last:
                ex.finish("Dt")
                self.dt_optimizer.step()
                self.dt_lr_scher.step((dt_loss_real + dt_loss_fake) if self._plateau else None)
        # ---------------- G, through the updated discriminators, weights of D held constant.  The D_t gradient exchange
        # of the last D iteration overlaps the D_s forward (D_s is already updated; D_t's update waits for the exchange)
        self._freeze_d(True)
        g_s_loss = self.calc_loss(self.D_s(fake_s, z_class), True)
        ex.finish("Dt")
        self.dt_optimizer.step()
        self.dt_lr_scher.step((dt_loss_real + dt_loss_fake) if self._plateau else None)
        g_t_loss = self.calc_loss(self.D_t(fake_d, z_class), True)
        self._freeze_d(False)
        self.g_optimizer.zero_grad()
        if ex.active:
            # bucketed exchange: the tail of the flat gradient buffer (last modules) is final first
            self._g_hi = self.g_optimizer.grad.numel()

            def on_ready(first_done, self=self, ex=ex):
                lo = self._g_bounds[first_done]
                # the bucket's weight gradients were queued on the side stream: the EXCHANGE stream waits for them -- the
                # step's chain is not fenced (round 6: joining the side stream here serialised the backward pass behind every
                # queued weight-gradient launch three times per step, bench.py --force-exchange)
                ex.start_range("G", self.g_optimizer.grad, lo, self._g_hi, after=(Fn.side_stream_if_any(),))
                self._g_hi = min(self._g_hi, lo)
            self.G.grad_ready_hook = on_ready
        (g_s_loss + g_t_loss).backward()
        Fn.join_side()
        if ex.active:
            self.G.grad_ready_hook = None
            ex.start_range("G", self.g_optimizer.grad, 0, self._g_hi)
        ex.finish("G")
        self.g_optimizer.step()
        self.g_lr_scher.step((g_s_loss + g_t_loss) if self._plateau else None)
        return ds_loss_real, ds_loss_fake, dt_loss_real, dt_loss_fake, g_s_loss, g_t_loss

    # ---- trainer.py:189-343 (loop; logging reduced to a print, no sampling)
    def _new_epoch(self):
        """-> iterator over the loader for the next epoch.  A rank-sharded loader (data.make_loader: DistributedSampler) shuffles
        with seed + epoch: without set_epoch every epoch would repeat the same order and the same rank split."""
        if hasattr(self.data_loader, "set_epoch"):
            self.data_loader.set_epoch(self._epoch)
        elif hasattr(getattr(self.data_loader, "sampler", None), "set_epoch"):
            self.data_loader.sampler.set_epoch(self._epoch)
        self._epoch += 1
        return iter(self.data_loader)

    def train(self):
        steps_per_epoch = len(self.data_loader)
        total_step = self.total_epoch * steps_per_epoch
        start = (self.pretrained_model + 1) if self.pretrained_model else 1
        self._epoch = (start - 1) // max(1, steps_per_epoch)       # a resumed run continues with the epoch it stopped in
        data_iter = self._new_epoch()
        self.D_s.train(); self.D_t.train(); self.G.train()
        t0 = time.time()
        for step in range(start, total_step + 1):
            try:
                real_videos, real_labels = next(data_iter)
            except StopIteration:
                data_iter = self._new_epoch()
                real_videos, real_labels = next(data_iter)
            losses = self.train_step(real_videos, real_labels)
            if self.log_epoch and step % (self.log_epoch * steps_per_epoch) == 0:
                vals = [float(v.detach()) for v in losses]
                print("Step: [%d/%d], time: %.1fs, ds_loss: %.4f, dt_loss: %.4f, g_s_loss: %.4f, g_t_loss: %.4f, lr: %.2e"
                      % (step, total_step, time.time() - t0, vals[0] + vals[1], vals[2] + vals[3], vals[4], vals[5],
                         self.g_lr_scher.get_lr()[0]))
            if self.model_save_epoch and step % (self.model_save_epoch * steps_per_epoch) == 0:
                self.save_models(step)

    # ---- trainer.py:323-334: the sampling path (eval-mode G on fixed z / labels, BN running statistics), without
    # the image-file side (torchvision save_image / tensorboard are host plumbing)
    @torch.no_grad()
    def sample(self, fixed_z, fixed_label):
        """-> denorm(G(fixed_z, fixed_label)) [B, T, 3, H, W] in [0, 1]; G is put back in train mode, like the reference.
        Note quirk 2: the spectral-norm u/v of G advance in eval mode as well."""
        self.G.eval()
        fake = self.G(fixed_z.to(self.device), fixed_label.to(self.device))
        self.G.train()
        return denorm(fake)

    # ---- trainer.py:337-343 / 375-382: reference-compatible checkpoints
    def save_models(self, step):
        """Rank 0 writes (its batch-norm running statistics are the ones saved in "replica" mode); the others wait."""
        world = D.world_size()
        if world == 1 or torch.distributed.get_rank() == 0:
            os.makedirs(self.model_save_path, exist_ok=True)
            for net, tag in ((self.G, "G"), (self.D_s, "Ds"), (self.D_t, "Dt")):
                torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()},
                           os.path.join(self.model_save_path, "{}_{}.pth".format(step, tag)))
        if world > 1:
            torch.distributed.barrier()

    def load_pretrained_model(self):
        for net, tag in ((self.G, "G"), (self.D_s, "Ds"), (self.D_t, "Dt")):
            sd = torch.load(os.path.join(self.model_save_path, "{}_{}.pth".format(self.pretrained_model, tag)),
                            map_location="cpu")
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}     # DataParallel prefix
            net.load_state_dict(sd)
