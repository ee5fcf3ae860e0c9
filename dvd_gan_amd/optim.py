"""Adam on flat fp32 buffers: one kernel launch per network per step (trainer.py:136-141 uses
torch.optim.Adam(lr, betas=(beta1, beta2)) on the requires_grad parameters).

The trainable parameters of a network are re-homed as views into ONE contiguous buffer, their
.grad tensors are views into a second one.  That makes the optimizer a single elementwise launch
and the data-parallel gradient exchange a single RCCL all-reduce per network.
"""
import torch

from . import kern as K


class FlatAdam:
    def __init__(self, params, lr, betas=(0.0, 0.9), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            off += k
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.param_groups = [{"lr": lr}]          # so torch lr schedulers' arithmetic can be mirrored

    def zero_grad(self):
        self.grad.zero_()                         # .grad tensors are views of this buffer

    def rebind(self):
        """Restore the .grad views if foreign code replaced them (e.g. zero_grad(set_to_none=True))."""
        off = 0
        for p in self.params:
            k = p.numel()
            view = self.grad[off:off + k].view(p.shape)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += k

    def step(self):
        self.t += 1
        K.adam_step(self.flat, self.grad, self.m, self.v, self.param_groups[0]["lr"], self.betas[0], self.betas[1],
                    self.eps, self.t)
