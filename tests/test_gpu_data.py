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
