"""Real-data input path (dvd_gan_amd/data.py) against the reference's UCF101 dataset + training transforms (fixture F12: a
tiny synthetic JPEG folder and the clips the reference produced from it under seeded `random`): host-side index / decode /
crop-resize, device-side flip + normalise + layout.  Bit-exact: the pixels come from the same PIL calls, the device kernel
does one division, one subtraction and one division per value like ToTensor + Normalize."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _materialise(g, root):
    for rel in [str(x) for x in g["meta.files"]]:
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(g["file." + rel].tobytes())


def test_reader_reproduces_the_reference_clips(golden, tmp_path):
    from dvd_gan_amd import data as D
    g = golden("f12_ucf101_reader")
    root = str(tmp_path)
    _materialise(g, root)
    sets = {m: D.UCF101(os.path.join(root, "jpg"), os.path.join(root, "ucf101_01.json"), "training", n_frames=8,
                        sample_size=16, train_crop=m) for m in ("corner", "random", "center")}
    assert len(sets["corner"]) == 3 and sets["corner"].class_names == {0: "ApplyEyeMakeup", 1: "Archery"}
    for tag in [str(x) for x in g["meta.cases"]]:
        mode, index, seed = tag.split(".")
        random.seed(100 * int(seed) + int(index))
        clip, flip, label = sets[mode][int(index)]
        assert label == int(g["out.label." + tag])
        got = D.clips_to_device(clip[None], flip[None], torch.device("cuda"))[0].cpu().numpy()
        want = g["out.clip." + tag]
        assert got.shape == want.shape, (tag, got.shape, want.shape)
        np.testing.assert_array_equal(got, want, err_msg=tag)


def test_loader_feeds_the_trainer_shape(golden, tmp_path):
    """make_loader -> (fp32 [B,3,T,S,S] on the device in [-1,1], labels): the tuple Trainer.train consumes."""
    from dvd_gan_amd import data as D
    g = golden("f12_ucf101_reader")
    _materialise(g, str(tmp_path))
    ds = D.UCF101(os.path.join(str(tmp_path), "jpg"), os.path.join(str(tmp_path), "ucf101_01.json"), n_frames=8, sample_size=16)
    random.seed(0)
    batches = list(D.make_loader(ds, batch_size=2, shuffle=False))
    assert len(batches) == 1
    clips, labels = batches[0]
    assert clips.is_cuda and clips.dtype == torch.float32 and tuple(clips.shape) == (2, 3, 8, 16, 16)
    assert float(clips.min()) >= -1.0 and float(clips.max()) <= 1.0 and labels.tolist() == [0, 1]


def test_trainer_train_loop_on_the_reader(golden, tmp_path):
    """End to end: the JPEG-folder reader feeds `Trainer.train()` (trainer.py:189-343 loop: epochs -> steps, loader re-iterated
    when exhausted, checkpoint written at the save interval) for two steps at 64x64, T=8, ch=2; parameters move, losses stay
    finite, the checkpoint has the reference's file names and keys."""
    import argparse
    from dvd_gan_amd import data as D
    from dvd_gan_amd.train_step import Trainer
    g = golden("f12_ucf101_reader")
    _materialise(g, str(tmp_path))
    ds = D.UCF101(os.path.join(str(tmp_path), "jpg"), os.path.join(str(tmp_path), "ucf101_01.json"), n_frames=8, sample_size=64)
    random.seed(4)
    torch.manual_seed(4)
    loader = D.make_loader(ds, batch_size=2, shuffle=False)
    assert len(loader) == 1
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=16, g_chn=2, ds_chn=2, dt_chn=2, n_frames=8, lr_schr="const",
                             total_epoch=2, d_iters=1, batch_size=2, g_lr=2e-3, d_lr=2e-3, beta1=0.0, beta2=0.9,
                             n_class=2, k_sample=4, model_save_path=str(tmp_path / "models"), version="t", model_save_epoch=2,
                             log_epoch=1)
    tr = Trainer(loader, cfg, device=torch.device("cuda"), compute_dtype=torch.float32)
    before = {k: v.detach().clone() for k, v in tr.G.state_dict().items()}
    tr.train()                                                  # 2 epochs x 1 step
    torch.cuda.synchronize()
    after = tr.G.state_dict()
    moved = [k for k in before if before[k].is_floating_point() and not torch.equal(before[k], after[k])]
    assert "conv.0.cells.0.update_gate.weight" in moved and "colorize.module.weight_u" in moved
    assert all(torch.isfinite(v).all() for v in after.values() if v.is_floating_point())
    for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):
        sd = torch.load(os.path.join(str(tmp_path / "models"), "t", f"2_{tag}.pth"))
        assert list(sd) == list(net.state_dict())
