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


def test_device_batcher_coco_shapes_feed_the_fused_step():
    """Colour images (N,3,32,32) uint8 + fp32 caption vectors: the async H2D pipeline of the COCO configuration."""
    from multimodal_vae_amd import data as D
    from multimodal_vae_amd import coco as M
    dev = _dev()
    T, B = 6, 8
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (40, 3, 32, 32), dtype=torch.uint8, generator=g)
    t = 0.4 * torch.randn(40, T, 300, generator=g)
    loader = D.DeviceBatcher(x, t, B, dev, shuffle=False)
    vae = M.MultimodalVAE(100, use_cuda=True, sos=0.4 * torch.randn(300, generator=g), steps=T).cuda()
    tr = M.FusedTrainer(vae, B, lr=1e-3)
    for b, (img, txt) in enumerate(loader):
        assert img.shape == (B, 3, 32, 32) and txt.shape == (B, T, 300) and txt.dtype == torch.float32
        assert torch.equal(img.cpu(), x[b * B:(b + 1) * B].float().div(255.0))
        assert torch.equal(txt.cpu(), t[b * B:(b + 1) * B])
        losses = tr(img, txt).losses()
    assert len(loader) == 5 and torch.isfinite(losses).all()


def test_device_batcher_device_gather_equals_host_gather():
    """The batch assembled by the GPU out of the pinned dataset (mmvae_gather_rows) == the host-side gather, shuffled."""
    from multimodal_vae_amd import data as D
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, 256, (96, 3, 32, 32), dtype=torch.uint8, generator=g)
    t = torch.randn(96, 102, 300, generator=g)
    a = D.DeviceBatcher(x, t, 16, dev, shuffle=True, seed=4, pin_dataset=True)
    b = D.DeviceBatcher(x, t, 16, dev, shuffle=True, seed=4, pin_dataset=False)
    assert a.device_gather and not b.device_gather
    n = 0
    for _ in range(2):                                         # two epochs: slot reuse
        for (ia, ta), (ib, tb) in zip(a, b):
            assert torch.equal(ia, ib) and torch.equal(ta, tb)
            n += 1
    assert n == 12


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


def test_coco_driver_runs_and_writes_reference_checkpoints(tmp_path):
    from multimodal_vae_amd import train_coco as T
    _dev()
    hist = T.main(["--cuda", "--epochs", "2", "--synthetic", "192", "--batch_size", "32", "--lr", "1e-3", "--log_interval", "2",
                   "--out", str(tmp_path / "ck"), "--results", str(tmp_path / "res")])
    assert len(hist["train"]) == 2 and all(np.isfinite(v) for e in hist["train"] + hist["test"] for v in e)
    assert sum(hist["train"][1]) < sum(hist["train"][0])
    ck = torch.load(tmp_path / "ck" / "checkpoint.pth.tar", weights_only=False)
    assert set(ck) == {'state_dict', 'best_loss', 'joint_loss', 'image_loss', 'text_loss', 'n_latents', 'optimizer'}   # coco/train.py:240-248
    assert "text_decoder.gru.weight_ih_l1" in ck['state_dict'] and ck['state_dict']["text_decoder.h2o.weight"].shape == (300, 300)
    assert torch.load(tmp_path / "res" / "sample_text_vector.pt").shape == (64, 102, 300)                             # :262-263


def test_eval_consumers_match_oracle():
    """compute_nll (multimnist/loglikelihood.py:20-69) and test_multimnist (multimnist/test.py:23-59) on the HIP modules."""
    from multimodal_vae_amd import multimnist as M
    from multimodal_vae_amd.evaluate import compute_nll, test_multimnist
    from oracle import mmvae_ref as R
    import torch.nn.functional as F
    dev = _dev()
    D, B = 100, 16
    P = R.formula_params("multimnist", D)
    vae = M.MultimodalVAE(D, use_cuda=True)
    vae.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    vae.cuda()
    image, text = R.formula_inputs("multimnist", 2 * B)
    loader = [(image[:B], text[:B]), (image[B:], text[B:])]
    for kw in (dict(), dict(image_only=True), dict(text_only=True)):
        torch.manual_seed(11)
        got = compute_nll(vae, loader, n_samples=2, use_cuda=True, **kw)
        torch.manual_seed(11)
        wi, wt = 0.0, 0.0
        with torch.no_grad():
            for im, tx in loader:
                a = None if kw.get("text_only") else im
                b = None if kw.get("image_only") else tx
                _, _, mu, lv, _ = R.multimnist_forward(P, a, b, False)
                sample = torch.randn(2, D)
                z = sample.unsqueeze(0) * lv.mul(0.5).exp().unsqueeze(1) + mu.unsqueeze(1)
                for i in range(2):
                    ri = R.multimnist_image_decoder(P, z[:, i], False)
                    rt, _ = R.multimnist_text_decoder(P, z[:, i], False)
                    wi += F.binary_cross_entropy(ri, im, reduction="sum").item() / 2
                    wt += F.nll_loss(rt.reshape(-1, 12), tx.reshape(-1), reduction="sum").item() / 2
        np.testing.assert_allclose(got, (wi / (2 * B), wt / (2 * B)), rtol=1e-2)
    acc = test_multimnist(vae, loader, use_cuda=True, verbose=False)
    with torch.no_grad():
        cc = lc = 0.0
        for im, tx in loader:
            _, rt, _, _, _ = R.multimnist_forward(P, im, None, False)
            pred = rt.argmax(2).numpy(); gt = tx.numpy()
            cc += (pred == gt).sum(); lc += ((pred == 11).sum(1) == (gt == 11).sum(1)).sum()
    assert abs(acc[0] - cc / (2 * B * 4)) <= 0.05 and abs(acc[1] - lc / (2 * B)) <= 0.1      # bf16 near-ties may flip an argmax


def test_coco_loss_variant_matches_torch():
    """coco/train.py:66-84: image BCE over 3*32*32 + F.mse_loss on the caption embeddings + KL/B*kl_lambda."""
    import torch.nn.functional as F
    from multimodal_vae_amd import coco as C
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    B, D = 6, 100
    mu = torch.randn(B, D, generator=g); lv = 0.3 * torch.randn(B, D, generator=g)
    ri = torch.rand(B, 3, 32, 32, generator=g).clamp(1e-4, 1 - 1e-4); im = torch.rand(B, 3, 32, 32, generator=g)
    rt = torch.randn(B, 102, 300, generator=g); tx = 0.4 * torch.randn(B, 102, 300, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (mu, lv, ri, rt)]
    ref = (0.7 * F.binary_cross_entropy(ref_in[2].view(-1, 3072), im.view(-1, 3072)) + 1.3 * F.mse_loss(ref_in[3], tx)
           + (-0.5 * torch.sum(1 + ref_in[1] - ref_in[0].pow(2) - ref_in[1].exp())) / B * 1e-2)
    ref.backward()
    got_in = [t.clone().to(dev).requires_grad_(True) for t in (mu, lv, ri, rt)]
    got = C.loss_function(got_in[0], got_in[1], recon_image=got_in[2], image=im.to(dev), recon_text=got_in[3], text=tx.to(dev),
                          kl_lambda=1e-2, lambda_xy=0.7, lambda_yx=1.3)
    got.backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=1e-5)
    for a, b in zip(got_in, ref_in):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-8)
    with pytest.raises(C.MMVAEError):
        C.MultimodalVAE(100)


def test_celeba_eval_consumers_match_oracle(tmp_path):
    """compute_nll (celeba/loglikelihood.py:18-67) and the three eval-mode losses of celeba/test.py:44-71 on the HIP modules, and
    the sample.py script entry (multimnist/sample.py:59-144) end to end on a checkpoint."""
    from multimodal_vae_amd import celeba as M
    from multimodal_vae_amd.evaluate import compute_nll_celeba, test_celeba
    from oracle import mmvae_ref as R
    import torch.nn.functional as F
    dev = _dev()
    D, B = 100, 8
    P = R.formula_params("celeba", D)
    vae = M.MultimodalVAE(D, use_cuda=True)
    vae.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    vae.cuda()
    image, attrs = R.formula_inputs("celeba", 2 * B)
    loader = [(image[:B], attrs[:B]), (image[B:], attrs[B:])]

    def attrs_dec(z):                                          # celeba/model.py:181-196 in eval mode (oracle pieces)
        x = F.linear(z, P["attrs_decoder.net.0.weight"], P["attrs_decoder.net.0.bias"])
        x = R.swish(R.batch_norm(x, P, "attrs_decoder.net.1", False))
        return torch.sigmoid(F.linear(x, P["attrs_decoder.net.3.weight"], P["attrs_decoder.net.3.bias"]))
    for kw in (dict(), dict(image_only=True), dict(attrs_only=True)):
        torch.manual_seed(5)
        got = compute_nll_celeba(vae, loader, n_samples=2, use_cuda=True, **kw)
        torch.manual_seed(5)
        wi, wa = 0.0, 0.0
        with torch.no_grad():
            for im, at in loader:
                a = None if kw.get("attrs_only") else im
                b = None if kw.get("image_only") else at
                _, _, mu, lv = R.celeba_forward(P, a, b, False)[:4]
                sample = torch.randn(2, D)
                z = sample.unsqueeze(0) * lv.mul(0.5).exp().unsqueeze(1) + mu.unsqueeze(1)
                for i in range(2):
                    wi += F.binary_cross_entropy(R.celeba_image_decoder(P, z[:, i], False), im, reduction="sum").item() / 2
                    wa += F.binary_cross_entropy(attrs_dec(z[:, i]), at, reduction="sum").item() / 2
        np.testing.assert_allclose(got, (wi / (2 * B), wa / (2 * B)), rtol=1e-2)
    got = test_celeba(vae, loader, use_cuda=True, verbose=False)
    want = [0.0, 0.0, 0.0]
    with torch.no_grad():
        for im, at in loader:
            for k, (a, b) in enumerate(((im, at), (im, None), (None, at))):
                ri, ra, mu, lv = R.celeba_forward(P, a, b, False)[:4]
                want[k] += float(R.celeba_loss(mu, lv, ri, im, ra, at, 1.0, 1.0, 1.0)) / 2
    np.testing.assert_allclose(got, want, rtol=2e-3)


def test_sample_script_entry(tmp_path):
    from multimodal_vae_amd import multimnist as M
    from multimodal_vae_amd.evaluate import _main
    from multimodal_vae_amd.train import save_checkpoint
    from oracle import mmvae_ref as R
    _dev()
    P = R.formula_params("multimnist", 100)
    vae = M.MultimodalVAE(100, use_cuda=True)
    vae.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    save_checkpoint({'state_dict': vae.state_dict(), 'best_loss': 0.0, 'n_latents': 100, 'optimizer': {}}, False, folder=str(tmp_path))
    img = (R.formula_inputs("multimnist", 1)[0][0, 0] * 255).round().to(torch.uint8)
    torch.save(img, tmp_path / "cond.pt")
    for extra in ([], ["--condition_on_image", str(tmp_path / "cond.pt")], ["--condition_on_text", "42"],
                  ["--condition_on_image", str(tmp_path / "cond.pt"), "--condition_on_text", "42"]):
        _main(["sample", str(tmp_path / "checkpoint.pth.tar"), "--n_samples", "8", "--out", str(tmp_path / "res")] + extra)
        im = torch.load(tmp_path / "res" / "sample_image.pt")
        assert im.shape == (8, 1, 50, 50) and float(im.min()) >= 0.0 and float(im.max()) <= 1.0
        lines = open(tmp_path / "res" / "sample_text.txt").read().splitlines()
        assert len(lines) == 8 and all(len(l) <= 4 and (l == "" or l.replace("^", "").isdigit() or "^" in l) for l in lines)
