"""COCO loss variant of the ELBO (SURVEY §8 a9'): ``loss_function`` of ``coco/train.py:66-84`` -- image BCE over
3*32*32 pixels, ``F.mse_loss`` on the GloVe caption embeddings, KL / B * kl_lambda -- on the HIP loss kernels.

The COCO *model* (coco/model.py: 32x32 conv stack + GloVe GRU caption encoder/decoder over 102 words) is not built in
this round; constructing it raises.
"""
from __future__ import annotations

import torch

from ._lib import MMVAEError, call, ptr
from .multimnist import _BCEMeanFn, _KLSumFn, _stream


class _MSEMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous().float(), b.contiguous().float()
        out = torch.zeros(1, dtype=torch.float32, device=a.device)
        call("mmvae_mse_fwd", ptr(a), ptr(b), a.numel(), ptr(out), _stream())
        ctx.save_for_backward(a, b)
        return (out / a.numel()).squeeze(0)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        call("mmvae_mse_bwd", ptr(a), ptr(b), a.numel(), float(g.item()) / a.numel(), ptr(da), _stream())
        return da, None


def loss_function(mu, logvar, recon_image=None, image=None, recon_text=None, text=None,
                  kl_lambda=1e-3, lambda_xy=1., lambda_yx=1.):
    """coco/train.py:66-84"""
    batch_size = mu.size(0)
    image_BCE, text_BCE = 0, 0
    if recon_image is not None and image is not None:
        image_BCE = lambda_xy * _BCEMeanFn.apply(recon_image.reshape(-1, 3 * 32 * 32), image.reshape(-1, 3 * 32 * 32))
    if recon_text is not None and text is not None:
        text_BCE = lambda_yx * _MSEMeanFn.apply(recon_text, text)
    KLD = _KLSumFn.apply(mu, logvar)
    KLD = KLD / batch_size * kl_lambda
    return image_BCE + text_BCE + KLD


elbo_loss = loss_function


class MultimodalVAE:
    def __init__(self, *a, **kw):
        raise MMVAEError("the COCO model family (coco/model.py) is not built yet: only its loss_function is (DESIGN.md §8)")
