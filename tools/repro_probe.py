"""Run-to-run reproducibility of a bf16 training step: two Trainers built from the same seed take the same step (same clips,
same RNG draws); every parameter, every spectral-norm / batch-norm buffer and the six losses are compared BITWISE.
usage: python tools/repro_probe.py [ch] [T] [B] [steps]      (prints the tensors that differ)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd.train_step import Trainer


def build(ch, T, B, ncls, seed):
    torch.manual_seed(seed)
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=ch, ds_chn=ch, dt_chn=ch, n_frames=T, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=ncls, k_sample=min(8, T))
    return Trainer([], cfg, device=torch.device("cuda", 0), compute_dtype=torch.bfloat16)


GRADS = {}


def run(ch, T, B, ncls, steps):
    from dvd_gan_amd import optim
    tr = build(ch, T, B, ncls, 3)
    GRADS.clear()
    orig = optim.FlatAdam.step

    def step(self):                       # the gradients as the optimizer sees them (a sign flip of a ~0 gradient moves a parameter by 2 lr)
        for net, tag in ((tr.G, "G"), (tr.D_s, "Ds"), (tr.D_t, "Dt")):
            ps = [p for p in net.parameters() if p.requires_grad]
            if ps and ps[0] is self.params[0]:
                for (k, p) in net.named_parameters():
                    if p.requires_grad:
                        GRADS[f"grad{self.t}.{tag}.{k}"] = p.grad.detach().clone()
        return orig(self)
    optim.FlatAdam.step = step
    g = torch.Generator().manual_seed(11)
    losses = []
    for s in range(steps):
        real = torch.rand(B, 3, T, 64, 64, generator=g) * 2 - 1
        labels = torch.randint(0, ncls, (B,), generator=g)
        draws = {"perm_real": torch.randperm(T, generator=g), "z": torch.randn(B, 120, generator=g),
                 "z_class": torch.randint(0, ncls, (B,), generator=g), "perm_fake": torch.randperm(T, generator=g)}
        losses.append([float(v.detach()) for v in tr.train_step(real, labels, draws)])
    torch.cuda.synchronize()
    optim.FlatAdam.step = orig
    state = dict(GRADS)
    for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):
        for k, v in net.state_dict().items():
            state[tag + "." + k] = v.detach().clone()
    return losses, state


def main():
    ch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    a = run(ch, T, B, 7, steps)
    b = run(ch, T, B, 7, steps)
    print("losses equal:", a[0] == b[0], a[0][-1])
    bad = [(k, float((a[1][k].double() - b[1][k].double()).abs().max())) for k in a[1] if not torch.equal(a[1][k], b[1][k])]
    print(f"{len(bad)} of {len(a[1])} tensors differ")
    for k, d in bad[:400]:
        if k.startswith("grad"):
            print(f"  {k:60s} max |diff| {d:.3e}  rel {d / (float(a[1][k].abs().max()) + 1e-30):.2e}")


if __name__ == "__main__":
    main()
