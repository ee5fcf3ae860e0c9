"""utils.py:60-63 / :77-83 of the reference on HIP kernels (reference-layout fp32 tensors)."""
import torch

from . import functional as Fn


def draw_frame_ids(video_length, k_sample, generator=None):
    """`torch.randperm(T)[:k].sort()` on the CPU default generator, like the reference."""
    perm = torch.randperm(video_length, generator=generator)
    return perm[:k_sample].sort()[0]


def sample_k_frames(data, video_length, k_sample, frame_ids=None):
    """data [B,T,C,H,W] -> [B,k,C,H,W]; the same k sorted random frames for the whole batch."""
    if frame_ids is None:
        frame_ids = draw_frame_ids(video_length, k_sample)
    return Fn.GatherFrames.apply(data, frame_ids)


def vid_downsample(data):
    """[B,T,C,H,W] -> per-frame 2x2 average -> [B,C,T,H/2,W/2]"""
    return Fn.VidDownsample.apply(data)


def denorm(x):
    """utils.py:41-43: [-1, 1] -> [0, 1], clamped."""
    return ((x + 1) / 2).clamp_(0, 1)
