"""Edge cases of the batch dimension: ragged row tiles (batch sizes that are not multiples of any tile height), the
smallest trainable batch (2), and the BatchNorm error for a batch of 1 -- every family, fused step vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import mmvae_ref as R

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _load(st, P):
    for n, shape, off in st.table:
        st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(st.device)


def _total(P):
    return torch.sqrt(sum(p.grad.double().pow(2).sum() for p in P.values() if p.grad is not None)).item()


@pytest.mark.parametrize("B", [2, 7, 33])
def test_multimnist_ragged_batches(B):
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    dev = _dev()
    D = 100
    P = R.formula_params("multimnist", D, requires_grad=True)
    st = MultimnistState(D, dev); _load(st, P)
    image, text = R.formula_inputs("multimnist", B)
    eps = [R.formula_eps(B, D, k) for k in range(3)]
    # reference token path first (the greedy feedback is forced to it: bf16 near-ties may flip an argmax)
    with torch.no_grad():
        Pc = {k: v.detach().clone() for k, v in P.items()}
        _, outs = R.multimnist_step_losses(Pc, image, text, True, 1e-3, eps, None, None, None, 0.0, 0.0)
    ft = [o[1].argmax(2) for o in outs]
    eng = FusedELBOStep(st, B)
    eng.enc_dropout = eng.gru_dropout = False
    out = eng.forward_backward(image.to(dev), text.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                               force_tokens=torch.stack(ft).reshape(3 * B, 4).to(dev).contiguous())
    losses, _ = R.multimnist_step_losses(P, image, text, True, 1e-3, eps, None, None, ft, 0.0, 0.0)
    sum(losses).backward()
    np.testing.assert_allclose(out.losses().cpu().numpy(), np.array([l.item() for l in losses]), rtol=2e-3)
    np.testing.assert_allclose(st.grads.double().norm().item(), _total(P), rtol=3e-2)
    assert torch.isfinite(st.grads).all()


@pytest.mark.parametrize("B", [2, 5, 37])
def test_mnist_ragged_batches(B):
    from multimodal_vae_amd.core import FusedMnistStep, MnistState
    dev = _dev()
    D = 20
    P = R.formula_params("mnist", D, requires_grad=True)
    st = MnistState(D, dev); _load(st, P)
    image, label = R.formula_inputs("mnist", B)
    eps = [R.formula_eps(B, D, k) for k in range(3)]
    eng = FusedMnistStep(st, B)
    out = eng.forward_backward(image.reshape(B, 784).to(dev).contiguous(), label.to(dev), True, True,
                               eps=torch.stack(eps).to(dev).contiguous())
    losses, _ = R.mnist_step_losses(P, image, label, True, eps)
    sum(losses).backward()
    np.testing.assert_allclose(out.losses().cpu().numpy(), np.array([l.item() for l in losses]), rtol=5e-5)
    np.testing.assert_allclose(st.grads.double().norm().item(), _total(P), rtol=1e-3)      # fp32 plan


@pytest.mark.parametrize("B", [3, 5])      # (B=2 makes BatchNorm1d of the attribute MLPs degenerate: xhat = +-1, rstd = 2/|x1-x2|)
def test_celeba_ragged_batches(B):
    from multimodal_vae_amd.core import FusedCelebaStep, CelebaState
    dev = _dev()
    D = 100
    P = R.formula_params("celeba", D, requires_grad=True)
    st = CelebaState(D, dev); _load(st, P)
    image, attrs = R.formula_inputs("celeba", B)
    eps = [R.formula_eps(B, D, k) for k in range(3)]
    eng = FusedCelebaStep(st, B)
    eng.enc_dropout = False
    out = eng.forward_backward(image.to(dev).contiguous(), attrs.to(dev).contiguous(), True, True,
                               eps=torch.stack(eps).to(dev).contiguous())
    losses, _ = R.celeba_step_losses(P, image, attrs, True, eps, None, 0.0)
    sum(losses).backward()
    np.testing.assert_allclose(out.losses().cpu().numpy(), np.array([l.item() for l in losses]), rtol=2e-3)
    np.testing.assert_allclose(st.grads.double().norm().item(), _total(P), rtol=5e-2)
    assert torch.isfinite(st.grads).all()


def test_batch_of_one_raises_like_batchnorm():
    """nn.BatchNorm in train mode refuses a single value per channel; the attribute/label MLPs hit that at B=1."""
    from multimodal_vae_amd.core import FusedMnistStep, MnistState
    from multimodal_vae_amd import MMVAEError
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    st = MnistState(20, dev); default_init_(st, 0)
    eng = FusedMnistStep(st, 1)
    with pytest.raises(MMVAEError, match="more than 1 value per channel"):
        eng.forward_backward(torch.rand(1, 784, device=dev), torch.zeros(1, dtype=torch.long, device=dev), True, True)
    # eval mode (running statistics) works at B=1
    out = eng.forward_backward(torch.rand(1, 784, device=dev), torch.zeros(1, dtype=torch.long, device=dev), False, False)
    assert torch.isfinite(out.losses()).all()
