"""utils.py:60-63 / :77-83 of the reference on HIP kernels (reference-layout fp32 tensors)."""
import torch

from . import functional as Fn


def to_device_async(t, device):
    """Host tensor -> device without stalling the host: through a pinned staging block, non_blocking.  A plain `.to(device)`
    of pageable memory makes the host wait for everything queued on the current stream -- once per step that keeps the host
    from running ahead of the GPU, and every launch gap of the step's Python becomes GPU idle time.  (The pinned block comes
    from torch's caching host allocator, which reuses it only after the copy has executed.)"""
    if t.is_cuda or not torch.cuda.is_available():
        return t.to(device)
    if not t.is_pinned():
        if t.numel() * t.element_size() > (1 << 20):
            return t.to(device)           # a batch of clips: pin it in the loader (make_loader does); no extra host copy here
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


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
