"""GPU checks of the driver (N1) and the device input pipeline (N3)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_device_batcher_equals_totensor():
    from multimodal_vae_amd import data as D
    from multimodal_vae_amd.utils import charlist_tensor
    dev = _dev()
    x, y = D.synthetic_multimnist(300, seed=5)
    t = torch.stack([charlist_tensor(l) for l in y])
    loader = D.DeviceBatcher(x, t, 64, dev, shuffle=False)
    assert len(loader) == 4                                   # ragged tail dropped
    seen = 0
    for b, (img, txt) in enumerate(loader):
        assert img.shape == (64, 1, 50, 50) and img.dtype == torch.float32 and txt.shape == (64, 4)
        ref = x[b * 64:(b + 1) * 64].float().div(255.0).unsqueeze(1)          # torchvision ToTensor semantics
        assert torch.equal(img.cpu(), ref)
        assert torch.equal(txt.cpu(), t[b * 64:(b + 1) * 64])
        seen += 1
    assert seen == 4
    # shuffled epochs cover the same multiset of samples
    loader2 = D.DeviceBatcher(x[:256], t[:256], 64, dev, shuffle=True, seed=1)
    sums = sorted(float(img.sum()) for img, _ in loader2)
    assert abs(sum(sums) - float(x[:256].float().sum() / 255.0)) < 1e-1


def test_driver_trains_and_writes_reference_checkpoints(tmp_path):
    from multimodal_vae_amd import train as T
    from multimodal_vae_amd.multimnist import MultimodalVAE
    _dev()
    hist = T.main(["--cuda", "--synthetic", "1024", "--epochs", "3", "--batch_size", "128", "--n_latents", "100", "--log_interval", "4",
                   "--anneal_lr", "--out", str(tmp_path / "ckpt"), "--results", str(tmp_path / "res")])
    tr = [sum(h) for h in hist["train"]]
    assert np.isfinite(tr).all() and tr[-1] < tr[0]
    assert np.isfinite(np.array(hist["test"])).all()
    ck = torch.load(tmp_path / "ckpt" / "checkpoint.pth.tar", weights_only=False)
    assert set(ck) == {"state_dict", "best_loss", "joint_loss", "image_loss", "text_loss", "n_latents", "optimizer"}   # train.py:243-251
    assert os.path.exists(tmp_path / "ckpt" / "model_best.pth.tar")
    vae = T.load_checkpoint(str(tmp_path / "ckpt" / "checkpoint.pth.tar"), use_cuda=True)
    assert isinstance(vae, MultimodalVAE)
    opt = torch.optim.Adam(vae.parameters(), lr=1e-3)
    opt.load_state_dict(ck["optimizer"])                         # the flat Adam moments load into torch.optim.Adam
    assert len(opt.state_dict()["state"]) == 52
    assert os.path.exists(tmp_path / "res" / "sample_text_epoch3.txt")
