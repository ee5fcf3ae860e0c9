"""checkpoint interchange with the reference (trainer.py:337-343, 375-382).
Reference-format files ({step}_G.pth / _Ds.pth / _Dt.pth = torch.save(state_dict), optionally with the
nn.DataParallel 'module.' prefix) must load into the HIP-backed modules and round-trip unchanged.
Runs on CPU (parameters only, no kernels)."""
import argparse
import os

import torch

from conftest import sub


def test_reference_checkpoints_roundtrip(tmp_path, golden):
    from dvd_gan_amd.gen_net import Generator
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    from dvd_gan_amd.train_step import Trainer
    g = golden("f9_trainer_hinge")
    ref = {tag: {k: torch.as_tensor(v) for k, v in sub(g, tag + ".sd0").items()} for tag in ("G", "Ds", "Dt")}
    # what the reference would have written, G with the DataParallel prefix
    torch.save({"module." + k: v for k, v in ref["G"].items()}, tmp_path / "7_G.pth")
    torch.save(ref["Ds"], tmp_path / "7_Ds.pth")
    torch.save(ref["Dt"], tmp_path / "7_Dt.pth")
    tr = Trainer.__new__(Trainer)                    # no GPU here: build the three networks by hand
    tr.G, tr.D_s, tr.D_t = Generator(16, 4, 3, 2, 8), SpatialDiscriminator(2, 3), TemporalDiscriminator(2, 3)
    tr.model_save_path, tr.pretrained_model = str(tmp_path), 7
    tr.load_pretrained_model()
    for tag, net in (("G", tr.G), ("Ds", tr.D_s), ("Dt", tr.D_t)):
        sd = net.state_dict()
        assert set(sd) == set(ref[tag])
        for k, v in ref[tag].items():
            assert torch.equal(sd[k].cpu(), v), (tag, k)
    tr.save_models(8)                                # and back: plain reference keys, loadable by the reference
    for tag in ("G", "Ds", "Dt"):
        back = torch.load(os.path.join(str(tmp_path), f"8_{tag}.pth"))
        assert list(back) == list(dict(getattr(tr, {"G": "G", "Ds": "D_s", "Dt": "D_t"}[tag]).state_dict()))
        for k, v in ref[tag].items():
            assert torch.equal(back[k], v), (tag, k)
