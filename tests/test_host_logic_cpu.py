"""Host-side logic that needs no GPU: the learning-rate schedules of trainer.py:142-176 against torch's own
schedulers and frame sampling against utils.py:60-63 semantics."""
import torch


def test_lr_schedules_match_torch():
    from dvd_gan_amd.train_step import _StepLR
    from torch.optim.lr_scheduler import ExponentialLR, MultiStepLR, StepLR

    class _Opt:                      # what _StepLR needs from an optimizer
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    base = 5e-5
    for kind, make in (("const", lambda o: StepLR(o, step_size=10000, gamma=1)),
                       ("step", lambda o: StepLR(o, step_size=500, gamma=0.98)),
                       ("exp", lambda o: ExponentialLR(o, gamma=0.9999)),
                       ("multi", lambda o: MultiStepLR(o, [10000, 30000], gamma=0.3))):
        p = torch.nn.Parameter(torch.zeros(1))
        ref_opt = torch.optim.Adam([p], base, (0.0, 0.9))
        ref = make(ref_opt)
        mine = _StepLR(_Opt(base), kind, base)
        for n in range(1, 30100):
            ref_opt.step()
            ref.step()
            mine.step()
            if n in (1, 499, 500, 501, 1000, 9999, 10000, 10001, 29999, 30000, 30001):
                assert abs(mine.get_lr()[0] - ref.get_last_lr()[0]) <= 1e-12 + 1e-9 * base, (kind, n)


def test_frame_ids_are_sorted_prefix_of_a_permutation():
    from dvd_gan_amd.helpers import draw_frame_ids
    g = torch.Generator().manual_seed(3)
    ids = draw_frame_ids(48, 8, g)
    assert ids.numel() == 8 and torch.equal(ids, ids.sort()[0]) and ids.unique().numel() == 8
    g = torch.Generator().manual_seed(3)
    assert torch.equal(ids, torch.randperm(48, generator=g)[:8].sort()[0])          # utils.py:61-62
    assert torch.equal(draw_frame_ids(6, 9), torch.arange(6))                       # k > T: every frame
