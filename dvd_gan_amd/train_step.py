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
from .dist import GradExchange
from .disc_nets import SpatialDiscriminator, TemporalDiscriminator
from .gen_net import Generator
from .helpers import denorm, draw_frame_ids, sample_k_frames, vid_downsample
from .optim import FlatAdam


class _StepLR:
    """lr schedule arithmetic of trainer.py:142-176 ('const' | 'step' | 'exp' | 'multi')."""

    def __init__(self, opt, kind, base_lr):
        self.opt, self.kind, self.base, self.n = opt, kind, base_lr, 0

    def step(self):
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


class Trainer(object):
    def __init__(self, data_loader, config, device=None, compute_dtype=torch.bfloat16, latent_dim=4):
        self.data_loader = data_loader
        c = config
        self.adv_loss, self.z_dim = c.adv_loss, c.z_dim
        self.g_chn, self.ds_chn, self.dt_chn = c.g_chn, c.ds_chn, c.dt_chn
        self.n_frames, self.lr_schr = c.n_frames, c.lr_schr
        self.total_epoch, self.d_iters, self.batch_size = c.total_epoch, c.d_iters, c.batch_size
        self.g_lr, self.d_lr, self.beta1, self.beta2 = c.g_lr, c.d_lr, c.beta1, c.beta2
        self.n_class, self.k_sample = c.n_class, c.k_sample
        self.pretrained_model = getattr(c, "pretrained_model", None)
        self.model_save_path = os.path.join(getattr(c, "model_save_path", "./models"), getattr(c, "version", ""))
        self.model_save_epoch = getattr(c, "model_save_epoch", 0)
        self.log_epoch = getattr(c, "log_epoch", 1)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.compute_dtype, self.latent_dim = compute_dtype, latent_dim
        self.exchange = GradExchange()
        self.build_model()
        if self.pretrained_model:
            self.load_pretrained_model()

    # ---- trainer.py:345-366
    def build_model(self):
        dt = self.compute_dtype
        self.G = Generator(self.z_dim, self.latent_dim, self.n_class, self.g_chn, self.n_frames, compute_dtype=dt).to(self.device)
        self.D_s = SpatialDiscriminator(self.ds_chn, self.n_class, compute_dtype=dt).to(self.device)
        self.D_t = TemporalDiscriminator(self.dt_chn, self.n_class, compute_dtype=dt).to(self.device)
        self.select_opt_schr()

    # ---- trainer.py:134-176
    def select_opt_schr(self):
        betas = (self.beta1, self.beta2)
        self.g_optimizer = FlatAdam(self.G.parameters(), self.g_lr, betas)
        self.G.dp_hooks = self.exchange.world > 1
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
        if self.lr_schr not in ("const", "step", "exp", "multi"):
            raise NotImplementedError("lr_schr='reduce' (ReduceLROnPlateau) is not ported")
        self.g_lr_scher = _StepLR(self.g_optimizer, self.lr_schr, self.g_lr)
        self.ds_lr_scher = _StepLR(self.ds_optimizer, self.lr_schr, self.d_lr)
        self.dt_lr_scher = _StepLR(self.dt_optimizer, self.lr_schr, self.d_lr)

    # ---- trainer.py:114-121
    def calc_loss(self, x, real_flag):
        return Fn.AdvLoss.apply(x, bool(real_flag), self.adv_loss == "hinge")

    # ---- trainer.py:84-88
    def label_sample(self):
        return torch.randint(low=0, high=self.n_class, size=(self.batch_size,)).to(self.device)

    # ---- trainer.py:384-387
    def reset_grad(self):
        self.ds_optimizer.zero_grad()
        self.dt_optimizer.zero_grad()
        self.g_optimizer.zero_grad()

    def _freeze_d(self, flag):
        self.D_s._set_train_weights(not flag)
        self.D_t._set_train_weights(not flag)

    # ---- trainer.py:223-307, one iteration (d_iters D updates + one G update)
    def train_step(self, real_videos, real_labels, draws=None):
        """real_videos [B,3,T,H,W], real_labels [B].  `draws` (tests): dict with the reference's RNG
        draws perm_real / z / z_class / perm_fake.  Returns the six loss terms as device scalars:
        ds_real, ds_fake, dt_real, dt_fake, g_s, g_t."""
        real_videos = real_videos.to(self.device).permute(0, 2, 1, 3, 4).contiguous()
        real_labels = real_labels.to(self.device)
        T, k = self.n_frames, self.k_sample
        ex = self.exchange
        for _ in range(self.d_iters):
            ids_real = draw_frame_ids(T, k) if draws is None else torch.as_tensor(draws["perm_real"])[:k].sort()[0]
            real_s = sample_k_frames(real_videos, T, k, ids_real)
            z = (torch.randn(self.batch_size, self.z_dim) if draws is None else torch.as_tensor(draws["z"])).to(self.device)
            z_class = self.label_sample() if draws is None else torch.as_tensor(draws["z_class"]).to(self.device)
            ex.finish("G")
            fake_videos = self.G(z, z_class)
            ids_fake = draw_frame_ids(T, k) if draws is None else torch.as_tensor(draws["perm_fake"])[:k].sort()[0]
            fake_s = sample_k_frames(fake_videos, T, k, ids_fake)
            # ---------------- D_s
            ds_loss_real = self.calc_loss(self.D_s(real_s, real_labels), True)
            ds_loss_fake = self.calc_loss(self.D_s(fake_s.detach(), z_class), False)
            self.reset_grad()
            (ds_loss_real + ds_loss_fake).backward()
            ex.start("Ds", self.ds_optimizer.grad)
            # ---------------- D_t (its forward/backward overlaps the D_s gradient exchange)
            real_d, fake_d = vid_downsample(real_videos), vid_downsample(fake_videos)
            dt_loss_real = self.calc_loss(self.D_t(real_d, real_labels), True)
            dt_loss_fake = self.calc_loss(self.D_t(fake_d.detach(), z_class), False)
            ex.finish("Ds")
            self.ds_optimizer.step()
            self.ds_lr_scher.step()
            self.dt_optimizer.zero_grad()
            (dt_loss_real + dt_loss_fake).backward()
            ex.start("Dt", self.dt_optimizer.grad)
            ex.finish("Dt")
            self.dt_optimizer.step()
            self.dt_lr_scher.step()
        # ---------------- G, through the updated discriminators, weights of D held constant
        self._freeze_d(True)
        g_s_loss = self.calc_loss(self.D_s(fake_s, z_class), True)
        g_t_loss = self.calc_loss(self.D_t(fake_d, z_class), True)
        self._freeze_d(False)
        self.g_optimizer.zero_grad()
        if ex.world > 1:
            # bucketed exchange: the tail of the flat gradient buffer (last modules) is final first
            self._g_hi = self.g_optimizer.grad.numel()

            def on_ready(first_done, self=self, ex=ex):
                lo = self._g_bounds[first_done]
                ex.start_range("G", self.g_optimizer.grad, lo, self._g_hi)
                self._g_hi = min(self._g_hi, lo)
            self.G.grad_ready_hook = on_ready
        (g_s_loss + g_t_loss).backward()
        if ex.world > 1:
            self.G.grad_ready_hook = None
            ex.start_range("G", self.g_optimizer.grad, 0, self._g_hi)
        ex.finish("G")
        self.g_optimizer.step()
        self.g_lr_scher.step()
        return ds_loss_real, ds_loss_fake, dt_loss_real, dt_loss_fake, g_s_loss, g_t_loss

    # ---- trainer.py:189-343 (loop; logging reduced to a print, no sampling)
    def train(self):
        data_iter = iter(self.data_loader)
        steps_per_epoch = len(self.data_loader)
        total_step = self.total_epoch * steps_per_epoch
        start = (self.pretrained_model + 1) if self.pretrained_model else 1
        self.D_s.train(); self.D_t.train(); self.G.train()
        t0 = time.time()
        for step in range(start, total_step + 1):
            try:
                real_videos, real_labels = next(data_iter)
            except StopIteration:
                data_iter = iter(self.data_loader)
                real_videos, real_labels = next(data_iter)
            losses = self.train_step(real_videos, real_labels)
            if self.log_epoch and step % (self.log_epoch * steps_per_epoch) == 0:
                vals = [float(v) for v in losses]
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
        os.makedirs(self.model_save_path, exist_ok=True)
        for net, tag in ((self.G, "G"), (self.D_s, "Ds"), (self.D_t, "Dt")):
            torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()},
                       os.path.join(self.model_save_path, "{}_{}.pth".format(step, tag)))

    def load_pretrained_model(self):
        for net, tag in ((self.G, "G"), (self.D_s, "Ds"), (self.D_t, "Dt")):
            sd = torch.load(os.path.join(self.model_save_path, "{}_{}.pth".format(self.pretrained_model, tag)),
                            map_location="cpu")
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}     # DataParallel prefix
            net.load_state_dict(sd)
