"""Where do the ATen helper launches of a training step come from?  One step under a TorchDispatchMode that records, for every
aten fill / zero / copy / add / cat / clone ... issued from Python, the innermost dvd_gan_amd source line on the stack (ops the
autograd engine issues itself -- gradient accumulation on its worker thread -- have no Python frame and are counted apart with a
second mode installed inside a backward hook).
usage: python tools/small_ops_report.py [batch]   -> table: count per step, op, source line"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvd_gan_amd.train_step import Trainer

WATCH = ("fill", "zero", "copy", "add", "cat", "clone", "mul", "empty_strided", "_to_copy", "arange", "index", "sum", "div", "sub")


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func)
        if any(w in full for w in WATCH):
            dev = None
            for a in list(args) + list((kwargs or {}).values()):
                if isinstance(a, torch.Tensor):
                    dev = a.device.type
                    break
            if dev == "cuda" or dev is None:
                site = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "dvd_gan_amd" in fr.filename or fr.filename.endswith("bench.py"):
                        site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                        break
                self.agg[(full.replace("aten.", ""), site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                             total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9,
                             n_class=101, k_sample=8)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tr = Trainer([], cfg, device=dev)
    real = (torch.rand(B, 3, 48, 64, 64) * 2 - 1).to(dev)
    labels = torch.randint(0, 101, (B,)).to(dev)
    tr.register_label_buffer(labels)
    for _ in range(2):
        tr.train_step(real, labels)
    torch.cuda.synchronize()
    rec = Recorder()
    with rec:
        tr.train_step(real, labels)
    torch.cuda.synchronize()
    rows = sorted(rec.agg.items(), key=lambda kv: -kv[1])
    print("watched aten ops issued from Python in one step:", sum(rec.agg.values()))
    for (name, site), n in rows[:70]:
        print(f"{n:5d}  {name:34s} {site}")


if __name__ == "__main__":
    main()
