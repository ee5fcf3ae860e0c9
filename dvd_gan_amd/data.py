"""Real-data input path: the UCF-101 JPEG-folder reader of the reference (Dataloader/datasets/ucf101.py:82-199 with the
training transforms main.py:42-55 builds) feeding the HIP trainer.

Split of work:
  host (DataLoader workers): JSON annotation index (`make_dataset`, same records as the reference), JPEG decode, temporal
      crop, the multi-scale crop + bilinear resize -- done with PIL exactly like the reference, so pixels are identical;
      a clip leaves the worker as uint8 [T, S, S, 3] plus its flip flag (4x smaller than the reference's fp32 tensor over
      PCIe);
  device (`clips_to_device`, kernel dvd_clip_to_tensor): horizontal flip, ToTensor(norm_value) + Normalize(mean, std) and the
      [T, S, S, 3] -> [3, T, S, S] layout change, one pass, into the fp32 [B, 3, T, S, S] batch `Trainer.train_step` takes.
Random draws come from Python's `random` in the reference's order (temporal crop, then scale, crop position, flip), so a seeded
run reproduces the reference's clips bit for bit (tests/test_gpu_data.py, fixture F12).
"""
import ctypes as C
import json
import math
import os
import random

import torch
import torch.utils.data as data

from . import lib as L


def load_value_file(path):                      # utils.py:54-58
    with open(path, "r") as f:
        return float(f.read().rstrip("\n\r"))


def make_dataset(root_path, annotation_path, subset, n_samples_for_each_video=1, sample_duration=16):
    """Index of the clips to draw from: one record per video of `subset`, or per `sample_duration`-frame window when
    n_samples_for_each_video > 1 -- the records Dataloader/datasets/ucf101.py:82-140 builds (same order, same keys), so a
    seeded run picks the same videos as the reference.  -> (records, {class index: class name})."""
    with open(annotation_path, "r") as f:
        ann = json.load(f)
    index_of = {label: n for n, label in enumerate(ann["labels"])}
    records = []
    for vid, entry in ann["database"].items():                 # file order = the reference's order
        cls = entry["annotations"]["label"]
        folder = os.path.join(root_path, cls, vid)
        if entry["subset"] != subset or not os.path.exists(folder) or cls not in index_of:
            continue
        total = int(load_value_file(os.path.join(folder, "n_frames")))
        if total <= 0:
            continue
        if n_samples_for_each_video == 1:
            windows = [(1, total + 1)]
        else:                                                   # windows spread evenly over the video, at least one frame apart;
            # n_samples_for_each_video < 1: back-to-back windows (ucf101.py:124-129: step = sample_duration)
            stride = (max(1, math.ceil((total - 1 - sample_duration) / (n_samples_for_each_video - 1)))
                      if n_samples_for_each_video > 1 else sample_duration)
            windows = [(first, min(total + 1, first + sample_duration)) for first in range(1, total, stride)]
        for lo, hi in windows:
            records.append({"video": folder, "segment": [1, total], "n_frames": total, "video_id": vid,
                            "label": index_of[cls], "frame_indices": list(range(lo, hi))})
    return records, {n: label for label, n in index_of.items()}


def temporal_random_crop(frame_indices, size):
    """`size` consecutive frames from a random start; a shorter clip is repeated cyclically up to `size`
    (Dataloader/transform/temporal_transforms.py:80-110; ONE random.randint draw, like the reference)."""
    first = random.randint(0, max(0, len(frame_indices) - size - 1))
    window = frame_indices[first:first + size]
    if not window or len(window) >= size:
        return list(window)
    return [window[i % len(window)] for i in range(size)]


class MultiScaleCrop:
    """MultiScaleCornerCrop / MultiScaleRandomCrop of Dataloader/transform/spatial_transforms.py:271-367 (crop box + PIL
    bilinear resize to `size`); `mode` = 'corner' | 'center' | 'random' as main.py:42-50 selects them."""

    def __init__(self, scales, size, mode="corner"):
        if mode not in ("corner", "center", "random"):
            raise ValueError("train_crop must be 'random', 'corner' or 'center'")
        self.scales, self.size, self.mode = scales, size, mode
        self.positions = ["c"] if mode == "center" else ["c", "tl", "tr", "bl", "br"]

    def randomize_parameters(self):
        self.scale = self.scales[random.randint(0, len(self.scales) - 1)]
        if self.mode == "random":
            self.tl_x, self.tl_y = random.random(), random.random()
        else:
            self.crop_position = self.positions[random.randint(0, len(self.positions) - 1)]

    def __call__(self, img):
        from PIL import Image
        w, h = img.size
        crop = int(min(w, h) * self.scale)
        if self.mode == "random":
            x1, y1 = self.tl_x * (w - crop), self.tl_y * (h - crop)
            box = (x1, y1, x1 + crop, y1 + crop)
        else:
            p = self.crop_position
            if p == "c":
                cx, cy, half = w // 2, h // 2, crop // 2
                box = (cx - half, cy - half, cx + half, cy + half)
            elif p == "tl":
                box = (0, 0, crop, crop)
            elif p == "tr":
                box = (w - crop, 0, w, crop)
            elif p == "bl":
                box = (0, h - crop, crop, h)
            else:
                box = (w - crop, h - crop, w, h)
        return img.crop(box).resize((self.size, self.size), Image.BILINEAR)


def default_scales(initial_scale=1.0, n_scales=5, scale_step=0.84089641525):     # main.py:38-40, parameter.py:73-75
    scales = [initial_scale]
    for _ in range(1, n_scales):
        scales.append(scales[-1] * scale_step)
    return scales


class UCF101(data.Dataset):
    """Training-set view of a UCF-101 JPEG folder (`<root>/<class>/<video>/image_%05d.jpg` + `n_frames`, annotation JSON of
    utils/ucf101_json.py).  __getitem__ -> (uint8 clip [T, S, S, 3], flip flag, class index)."""

    def __init__(self, root_path, annotation_path, subset="training", n_frames=16, sample_size=64, scales=None,
                 train_crop="corner", n_samples_for_each_video=1, sample_duration=16):
        self.data, self.class_names = make_dataset(root_path, annotation_path, subset, n_samples_for_each_video, sample_duration)
        self.n_frames = n_frames
        self.crop = MultiScaleCrop(scales or default_scales(), sample_size, train_crop)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        import numpy as np
        from PIL import Image
        rec = self.data[index]
        frame_indices = temporal_random_crop(list(rec["frame_indices"]), self.n_frames)
        self.crop.randomize_parameters()
        flip = random.random() < 0.5                   # RandomHorizontalFlip.randomize_parameters / __call__ (:253-268)
        frames = []
        for i in frame_indices:                        # video_loader (:37-46): stops at the first missing frame
            path = os.path.join(rec["video"], "image_{:05d}.jpg".format(i))
            if not os.path.exists(path):
                break
            with open(path, "rb") as f, Image.open(f) as img:
                frames.append(np.asarray(self.crop(img.convert("RGB")), dtype=np.uint8))
        clip = torch.from_numpy(np.stack(frames, 0))
        return clip, torch.tensor(bool(flip)), rec["label"]


def clips_to_device(clips, flips, device, norm_value=255.0, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """uint8 [B, T, S, S, 3] (host or device) + flip flags [B] -> fp32 [B, 3, T, S, S] on `device`:
    ((x / norm_value) - mean) / std with the flagged clips mirrored (main.py:34-36,51-54 defaults: [-1, 1])."""
    clips = clips.to(device, non_blocking=True).contiguous()
    flips = flips.to(device=device, dtype=torch.uint8).contiguous()
    B, T, S1, S2, ch = clips.shape
    assert ch == 3 and clips.dtype == torch.uint8
    out = torch.empty(B, 3, T, S1, S2, dtype=torch.float32, device=device)
    ms = torch.tensor(list(mean) + list(std), dtype=torch.float32, device=device)
    L.check(L.lib().dvd_clip_to_tensor(L.ptr(clips), L.ptr(flips), L.ptr(out), C.c_longlong(B), T, S1, S2,
                                       C.c_float(norm_value), L.ptr(ms), L.stream()))
    return out


def make_loader(dataset, batch_size, num_workers=0, shuffle=True, device=None, rank=None, world=None, seed=0):
    """DataLoader whose batches are (fp32 [B,3,T,S,S] on `device`, labels [B]) -- what Trainer.train expects.
    Two deliberate differences from the reference's DataLoader (main.py:57-62): the last, incomplete batch is DROPPED
    (the generator's batch size is fixed by z, and the ConvGRU / CBN buffers are sized for it), so len(loader) -- and with it
    steps_per_epoch / total_step / the log and save cadence of Trainer.train -- is floor(N / B), not ceil(N / B); and in a
    data-parallel run (rank / world given, or read from torch.distributed) every rank iterates ITS OWN 1/world of a common
    permutation (DistributedSampler seeded alike on all ranks; call `loader.set_epoch(e)` per epoch), so an epoch covers
    the data once instead of world times."""
    if world is None and torch.distributed.is_available() and torch.distributed.is_initialized():
        world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
    sampler = None
    if world is not None and world > 1:
        sampler = data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed,
                                                      drop_last=True)
    loader = data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                             num_workers=num_workers, pin_memory=True, drop_last=True)

    class _OnDevice:
        def __len__(self):
            return len(loader)

        def set_epoch(self, epoch):
            if sampler is not None:
                sampler.set_epoch(epoch)

        def __iter__(self):
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            for clips, flips, labels in loader:
                yield clips_to_device(clips, flips, dev), labels
    return _OnDevice()
