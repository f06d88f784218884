"""Weak-supervision step variants (multimnist/paired_weak.py:84-117, modal_weak.py:87-117, and the mnist
twins): a subset of the three passes per step.  An absent pass must contribute no loss, no gradient and no BatchNorm
running-statistics update -- checked against the oracle running exactly the passes the reference would run."""
import os

import numpy as np
import pytest
import torch

from oracle import mmvae_ref as R
from gradcheck import check_gradients

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _load(st, P):
    for n, shape, off in st.table:
        st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(st.device)


@pytest.mark.parametrize("case", ["unpaired", "joint_and_text"])
def test_multimnist_pass_subsets(case, golden_dir):
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    from multimodal_vae_amd import weak
    dev = _dev()
    D = 100
    fx = np.load(os.path.join(golden_dir, "multimnist_b8.npz"))
    B = int(fx["B"])
    P = R.formula_params("multimnist", D, requires_grad=True)
    st = MultimnistState(D, dev); _load(st, P)
    image, text = R.formula_inputs("multimnist", B)
    eps = [torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)]
    ft = [torch.from_numpy(fx[f"tokens_{k}"]).long() for k in range(3)]

    class _Rng:                                            # forces the branch of the reference's coin flips
        def __init__(self, vals): self.vals = list(vals)
        def random(self): return self.vals.pop(0)
    if case == "unpaired":                                 # paired_weak.py:104-113: only the two uni-modal passes
        cfg = weak.paired_weak(0.3, _Rng([0.9]))
        assert cfg["passes"] == (False, True, True)
    else:                                                  # modal_weak.py: joint always, image pass dropped, text pass kept
        cfg = weak.modal_weak(0.5, 0.5, _Rng([0.9, 0.1]))
        assert cfg["passes"] == (True, False, True)
    eng = FusedELBOStep(st, B)
    eng.enc_dropout = eng.gru_dropout = False
    out = eng.forward_backward(image.to(dev), text.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                               force_tokens=torch.stack(ft).reshape(3 * B, 4).to(dev).contiguous(), **cfg)
    # oracle: run exactly the present passes, with the loss terms the reference passes to loss_function
    args = ((image, text), (image, None), (None, text))
    total, want = 0, [0.0, 0.0, 0.0]
    for k in range(3):
        if not cfg["passes"][k]:
            continue
        ri, rt, mu, lv, _ = R.multimnist_forward(P, args[k][0], args[k][1], True, eps[k], None, None, ft[k], 0.0, 0.0)
        lxy, lyx = cfg["lambda_xy"][k], cfg["lambda_yx"][k]
        l = R.multimnist_loss(mu, lv, ri if lxy else None, image if lxy else None, rt if lyx else None, text if lyx else None,
                              1e-3, lxy or 1.0, lyx or 1.0)
        total = total + l
        want[k] = l.item()
    total.backward()
    np.testing.assert_allclose(out.losses().cpu().numpy(), np.array(want), rtol=1e-3, atol=1e-7)
    g = st.grads.cpu()
    check_gradients(((n, g[off:off + P[n].numel()], P[n].grad) for n, shape, off in st.table), 3.5e-2, None, "weak supervision")
    # BatchNorm running statistics: only the passes that exist updated them (oracle buffers were updated in place)
    for pre, c, off in st.bn_table:
        np.testing.assert_allclose(st.bn_stats[off:off + c].cpu().numpy(), P[pre + ".running_mean"].numpy(), atol=2e-3)
        np.testing.assert_allclose(st.bn_stats[off + c:off + 2 * c].cpu().numpy(), P[pre + ".running_var"].numpy(), rtol=2e-2, atol=1e-4)
    nbt = {pre: int(st.bn_nbt[i].item()) for i, (pre, _, _) in enumerate(st.bn_table)}
    for pre, v in nbt.items():
        assert v == int(P[pre + ".num_batches_tracked"].item()), pre


def test_mnist_unpaired_step():
    from multimodal_vae_amd.core import FusedMnistStep, MnistState
    dev = _dev()
    D, B = 20, 64
    P = R.formula_params("mnist", D, requires_grad=True)
    st = MnistState(D, dev); _load(st, P)
    image, label = R.formula_inputs("mnist", B)
    eps = [R.formula_eps(B, D, k) for k in range(3)]
    eng = FusedMnistStep(st, B)
    cfg = dict(passes=(False, True, True), lambda_xy=(0.0, 1.0, 0.0), lambda_yx=(0.0, 0.0, 1.0))     # mnist/paired_weak.py unpaired branch
    out = eng.forward_backward(image.reshape(B, 784).to(dev).contiguous(), label.to(dev), True, True,
                               eps=torch.stack(eps).to(dev).contiguous(), **cfg)
    ri, rt, mu, lv = R.mnist_forward(P, image, None, True, eps[1])          # default MNIST plan is fp32: fp32 tolerances
    l2 = R.mnist_loss(mu, lv, ri, image, None, None)
    ri, rt, mu, lv = R.mnist_forward(P, None, label, True, eps[2])
    l3 = R.mnist_loss(mu, lv, None, None, rt, label)
    (l2 + l3).backward()
    np.testing.assert_allclose(out.losses().cpu().numpy(), np.array([0.0, l2.item(), l3.item()]), rtol=2e-5, atol=1e-7)
    g = st.grads.cpu()
    tot = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in P.values() if p.grad is not None)).item()
    np.testing.assert_allclose(g.double().norm().item(), tot, rtol=1e-4)
    for i, (pre, c, off) in enumerate(st.bn_table):
        assert int(st.bn_nbt[i].item()) == int(P[pre + ".num_batches_tracked"].item()), pre
        np.testing.assert_allclose(st.bn_stats[off:off + c].cpu().numpy(), P[pre + ".running_mean"].numpy(), atol=1e-5)
