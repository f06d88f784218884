"""Eval-mode consumers of the model surface: the reference's ``compute_nll``
(multimnist/loglikelihood.py:20-69) and ``test_multimnist`` accuracy (multimnist/test.py:23-59), running on the HIP
modules in eval mode (BatchNorm running statistics, no dropout, z = mu).

``compute_nll`` quirk kept out: the reference calls ``F.nll_loss(recon_text (B,4,12), text (B,4), size_average=False)``
directly, which modern torch rejects (class dimension mismatch); the evident intent -- the summed negative log-likelihood
of the 4 target characters -- is what is computed here (``recon_text.view(-1, 12)`` against ``text.view(-1)``).
"""
from __future__ import annotations

import numpy as np
import torch

from .multimnist import _BCEMeanFn, _NLLMeanFn
from .utils import FILL, max_length


@torch.no_grad()
def compute_nll(model, loader, image_only=False, text_only=False, n_samples=1, use_cuda=True, verbose=False):
    """-> (image NLL per sample, text NLL per sample), multimnist/loglikelihood.py:20-69."""
    assert not (image_only and text_only)
    model.eval()
    test_image_nll, test_text_nll, n_seen = 0.0, 0.0, 0
    for batch_idx, (image, text) in enumerate(loader):
        if use_cuda:
            image, text = image.cuda(), text.cuda()
        if not image_only and not text_only:
            _, _, mu, logvar = model(image, text)
        elif image_only:
            _, _, mu, logvar = model(image=image)
        else:
            _, _, mu, logvar = model(text=text)
        batch_size, n_latents = mu.size(0), mu.size(1)
        sample = torch.randn(n_samples, n_latents)                # drawn on the host like the reference (:43)
        if use_cuda:
            sample = sample.cuda()
        std = logvar.mul(0.5).exp()
        z = sample.unsqueeze(0) * std.unsqueeze(1) + mu.unsqueeze(1)          # (B, n_samples, D)
        image_nll, text_nll = 0.0, 0.0
        for i in range(n_samples):
            zi = z[:, i].contiguous()
            recon_image = model.decode_image(zi)
            recon_text = model.decode_text(zi)
            image_nll += float(_BCEMeanFn.apply(recon_image.reshape(batch_size, -1), image.reshape(batch_size, -1))) * image.numel()
            text_nll += float(_NLLMeanFn.apply(recon_text.reshape(-1, recon_text.size(2)), text.reshape(-1))) * text.numel()
        test_image_nll += image_nll / n_samples
        test_text_nll += text_nll / n_samples
        n_seen += batch_size
        if verbose:
            print('Evaluating: [{}/{}]'.format(n_seen, len(loader) * batch_size))
    return test_image_nll / max(n_seen, 1), test_text_nll / max(n_seen, 1)


@torch.no_grad()
def test_multimnist(model, loader, use_cuda=True, verbose=True):
    """-> (character accuracy, length accuracy) of text predicted from the image alone, multimnist/test.py:23-59."""
    model.eval()
    char_correct, len_correct, n_seen = 0.0, 0.0, 0
    for image, text in loader:
        if use_cuda:
            image, text = image.cuda(), text.cuda()
        _, recon_text, _, _ = model(image=image)
        pred = torch.max(recon_text, dim=2)[1].cpu().numpy()
        gt = text.cpu().numpy()
        char_correct += float(np.sum(pred == gt))
        len_correct += float(np.sum(np.sum(pred == FILL, axis=1) == np.sum(gt == FILL, axis=1)))
        n_seen += len(gt)
    _char_correct = char_correct / max(n_seen * max_length, 1)
    _len_correct = len_correct / max(n_seen, 1)
    if verbose:
        print('\nTest set: Character Accuracy: {}/{} ({:.0f}%)\tLength Accuracy: {}/{} ({:.0f}%)\n'.format(
            int(char_correct), n_seen * max_length, 100. * _char_correct, int(len_correct), n_seen, 100. * _len_correct))
    return _char_correct, _len_correct


test_multimnist.__test__ = False          # not a pytest test
