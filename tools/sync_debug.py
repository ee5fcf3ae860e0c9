import argparse, os, sys, warnings, traceback
import torch
sys.path.insert(0, os.getcwd())
from dvd_gan_amd.train_step import Trainer
B=16
cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=32, ds_chn=32, dt_chn=32, n_frames=48, lr_schr="const",
                         total_epoch=1, d_iters=1, batch_size=B, g_lr=5e-5, d_lr=5e-5, beta1=0.0, beta2=0.9, n_class=101, k_sample=8)
dev = torch.device("cuda")
tr = Trainer([], cfg, device=dev)
real = (torch.rand(B, 3, 48, 64, 64) * 2 - 1).to(dev)
labels = torch.randint(0, 101, (B,)).to(dev)
tr.register_label_buffer(labels)
tr.train_step(real, labels); tr.train_step(real, labels)
torch.cuda.synchronize()
seen=set()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st=[f for f in traceback.extract_stack() if 'dvd_gan_amd' in f.filename]
    key=tuple((f.filename,f.lineno) for f in st[-3:])
    if key in seen: return
    seen.add(key)
    print("SYNC:", str(message)[:80]); 
    for f in st[-4:]: print("    %s:%d %s"%(os.path.basename(f.filename), f.lineno, f.line))
warnings.showwarning=showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr.train_step(real, labels)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("done")
