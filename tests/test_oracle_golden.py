"""CPU: the oracle restatement (oracle/mmvae_ref.py) reproduces the golden vectors that
oracle/make_golden.py captured from the imported reference (reference never needed here)."""
import os

import numpy as np
import pytest
import torch

from oracle import mmvae_ref as R


def _unpack(fx, B, widths):
    out = []
    for i in range(2):
        out.append([torch.from_numpy(np.unpackbits(fx[f"mask_{i}_{j}"], axis=1)[:, :w].astype(np.float32))
                    for j, w in enumerate(widths)])
    return out


def _run(ds, fx):
    B, D = int(fx["B"]), int(fx["D"])
    wm = bool(fx["with_masks"])
    P = R.formula_params(ds, D, requires_grad=True)
    image, second = R.formula_inputs(ds, B)
    full = "eps_0" in fx
    if full:
        eps = [torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)]
    else:  # scalar-only records were drawn with torch.manual_seed(seed0+k)
        eps = []
        for k in range(3):
            torch.manual_seed(int(fx["seed0"]) + k)
            eps.append(torch.empty(B, D).normal_())
    if ds == "multimnist":
        em = None
        if wm:
            m = _unpack(fx, B, (400, 200))
            em = (m[0], m[1], None)
        losses, outs = R.multimnist_step_losses(P, image, second, True, 1e-3, eps, em,
                                                enc_drop_p=0.1 if wm else 0.0, gru_drop_p=0.0)
    elif ds == "coco":
        em = None
        if wm:
            m = _unpack(fx, B, (1024, 256))
            em = (m[0], m[1], None)
        losses, outs = R.coco_step_losses(P, image, second, R.formula_sos(), True, 1e-3, eps, em,
                                          enc_drop_p=0.1 if wm else 0.0, gru_drop_p=0.0)
    elif ds == "mnist":
        losses, outs = R.mnist_step_losses(P, image.view(-1, 784), second, True, eps)
    else:
        em = None
        if wm:
            m = _unpack(fx, B, (1024,))
            em = (m[0][0], m[1][0], None)
        losses, outs = R.celeba_step_losses(P, image, second, True, eps, em, 0.1 if wm else 0.0)
    (losses[0] + losses[1] + losses[2]).backward()
    return P, losses, outs


CASES = [("multimnist", "multimnist_b8"), ("multimnist", "multimnist_b8_masks"), ("mnist", "mnist_b8"),
         ("mnist", "mnist_b128_scalars"), ("celeba", "celeba_b4"), ("celeba", "celeba_b4_masks"),
         ("coco", "coco_b4"), ("coco", "coco_b4_masks")]


@pytest.mark.parametrize("ds,name", CASES)
def test_oracle_matches_golden(ds, name, golden_dir):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    P, losses, outs = _run(ds, fx)
    D = int(fx["D"])
    names = [n for n, _ in R.param_table(ds, D)]
    np.testing.assert_allclose([l.item() for l in losses], fx["loss"], rtol=2e-5)
    gn = torch.sqrt(sum(P[n].grad.double().pow(2).sum() for n in names)).item()
    np.testing.assert_allclose(gn, float(fx["total_grad_norm"]), rtol=2e-4)
    if "mu_0" in fx:
        for k in range(3):
            np.testing.assert_allclose(outs[k][2].detach().numpy(), fx[f"mu_{k}"], atol=2e-5)
            np.testing.assert_allclose(outs[k][3].detach().numpy(), fx[f"logvar_{k}"], atol=2e-5)
            if ds == "coco":
                np.testing.assert_allclose(outs[k][1][:, :3].detach().numpy(), fx[f"text_recon_head_{k}"], atol=5e-5)
                np.testing.assert_allclose(outs[k][1][:, -2:].detach().numpy(), fx[f"text_recon_tail_{k}"], atol=5e-5)
                np.testing.assert_allclose(outs[k][1].detach().double().norm().item(), fx[f"text_recon_stats_{k}"][1], rtol=1e-5)
            else:
                np.testing.assert_allclose(outs[k][1].detach().numpy(), fx[f"second_recon_{k}"], atol=5e-5)
            s = outs[k][0].detach().double().reshape(-1)
            np.testing.assert_allclose(s.sum().item(), fx[f"image_recon_stats_{k}"][0], rtol=1e-5)
        gs = fx["grad_stats"]
        for i, n in enumerate(names):
            g = P[n].grad.double()
            np.testing.assert_allclose(g.norm().item(), gs[i, 1], rtol=5e-4, atol=1e-5 * gn)
        for key in fx.files:
            if key.startswith("buf:"):
                np.testing.assert_allclose(P[key[4:]].numpy(), fx[key], rtol=1e-4, atol=1e-6)
    else:
        np.testing.assert_allclose([P[n].grad.double().norm().item() for n in names], fx["grad_norms"],
                                   rtol=1e-3, atol=1e-5 * gn)


def load_replay_draws(fx, ds):
    """(eps[3], enc_masks[3], gru_masks[3]) of a *_dropout fixture (the reference's own draws, replayed by seed)."""
    B = int(fx["B"])
    widths = (400, 200) if ds == "multimnist" else (1024, 256)
    H = 100 if ds == "multimnist" else 200
    eps = [torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)]
    em, gm = [], []
    for k in range(3):
        if f"encmask_{k}_0" in fx.files:
            em.append([torch.from_numpy(np.unpackbits(fx[f"encmask_{k}_{j}"], axis=1)[:, :w].astype(np.float32)) for j, w in enumerate(widths)])
        else:
            em.append(None)
        g = np.unpackbits(fx[f"grukeep_{k}"], axis=2)[:, :, :H].astype(np.float32)      # (T, B, H)
        gm.append([torch.from_numpy(g[t]) for t in range(g.shape[0])])
    return eps, tuple(em), tuple(gm)


@pytest.mark.parametrize("ds,name", [("multimnist", "multimnist_b8_dropout"), ("multimnist", "multimnist_b256_dropout"),
                                     ("coco", "coco_b4_dropout")])
def test_oracle_matches_default_train_mode_fixture(ds, name, golden_dir):
    """Every Dropout at the reference's p = 0.1 (classifier AND nn.GRU inter-layer, multimnist/model.py:175,178,262), the
    reference's draws replayed by seed and stored: the oracle with those masks injected gives the reference's numbers."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    B, D = int(fx["B"]), int(fx["D"])
    P = R.formula_params(ds, D, requires_grad=True)
    image, second = R.formula_inputs(ds, B)
    eps, em, gm = load_replay_draws(fx, ds)
    for k in range(3):
        assert 0.8 < float(torch.stack(gm[k]).mean()) < 0.97            # keep rate of the stored masks ~ 0.9
    if ds == "multimnist":
        losses, outs = R.multimnist_step_losses(P, image, second, True, 1e-3, eps, em, gm, None, 0.1, 0.1)
    else:
        losses, outs = R.coco_step_losses(P, image, second, R.formula_sos(), True, 1e-3, eps, em, gm, 0.1, 0.1)
    (losses[0] + losses[1] + losses[2]).backward()
    names = [n for n, _ in R.param_table(ds, D)]
    np.testing.assert_allclose([l.item() for l in losses], fx["loss"], rtol=2e-5)
    gn = torch.sqrt(sum(P[n].grad.double().pow(2).sum() for n in names)).item()
    np.testing.assert_allclose(gn, float(fx["total_grad_norm"]), rtol=2e-4)
    for i, n in enumerate(names):
        g = P[n].grad.double().reshape(-1)
        idx = R.sample_idx(g.numel(), 64)
        np.testing.assert_allclose(g[idx].numpy(), fx["grad_samples"][i, :len(idx)], rtol=2e-3, atol=2e-6 * gn)
    if ds == "multimnist":
        for k in range(3):
            np.testing.assert_array_equal(outs[k][4].numpy(), fx[f"tokens_{k}"])     # greedy path of the reference


@pytest.mark.parametrize("ds,name", [("multimnist", "multimnist_b256_scalars"), ("celeba", "celeba_b512_scalars"),
                                     ("coco", "coco_b128_scalars")])
def test_oracle_matches_full_size_gradient_samples(ds, name, golden_dir):
    """BASELINE.json's configurations 2 / 3 / 5 (per-GPU share): gradient samples (direction) and BatchNorm buffers."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    P, losses, outs = _run(ds, fx)
    names = [n for n, _ in R.param_table(ds, int(fx["D"]))]
    np.testing.assert_allclose([l.item() for l in losses], fx["loss"], rtol=2e-5)
    gn = float(fx["total_grad_norm"])
    for i, n in enumerate(names):
        g = P[n].grad.double().reshape(-1)
        idx = R.sample_idx(g.numel(), 64)
        np.testing.assert_allclose(g[idx].numpy(), fx["grad_samples"][i, :len(idx)], rtol=5e-3, atol=5e-6 * gn)
    for key in fx.files:
        if key.startswith("buf:"):
            np.testing.assert_allclose(P[key[4:]].numpy(), fx[key], rtol=1e-4, atol=1e-6)
    if ds == "multimnist":
        for k in range(3):
            np.testing.assert_array_equal(outs[k][4].numpy(), fx[f"tokens_{k}"])
            assert float(fx[f"margin_min_{k}"]) > 0


def test_tokens_and_margins_recorded(golden_dir):
    fx = np.load(os.path.join(golden_dir, "multimnist_b8.npz"))
    assert fx["tokens_0"].shape == (8, 4)
    assert float(fx["margin_0"].min()) > 0


def test_param_table_counts():
    # SURVEY 8 a10/a12/a13 [probed]: 2,338,392 / 806,604 / 8,808,802 parameters
    def count(ds, D):
        return sum(int(np.prod(s)) for _, s in R.param_table(ds, D))
    assert count("multimnist", 100) == 2338392
    assert len(R.param_table("multimnist", 100)) == 52
    assert count("mnist", 20) == 806604
    assert count("celeba", 100) == 8808802
    assert count("coco", 100) == 9488180          # 9.49 M


def test_poe_is_variance_weighted():
    # SURVEY 2.2 K9 [probed]: experts mu=(0,1), var=(1,3) -> mu=0.75
    mu = torch.tensor([[[0.0]], [[1.0]]])
    lv = torch.log(torch.tensor([[[1.0]], [[3.0]]]))
    m, l = R.product_of_experts(mu, lv)
    assert abs(m.item() - 0.75) < 1e-6
    assert abs(l.exp().item() - 0.75) < 1e-6
