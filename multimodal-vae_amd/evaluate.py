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


# ------------------------------------------------------------------------------------------------------------------
# CelebA consumers (celeba/loglikelihood.py:18-67, celeba/test.py:44-71)

@torch.no_grad()
def compute_nll_celeba(model, loader, image_only=False, attrs_only=False, n_samples=1, use_cuda=True, verbose=False):
    """-> (image NLL per sample, attribute NLL per sample), celeba/loglikelihood.py:18-67.

    Quirk kept out: the reference scores the attributes with ``F.nll_loss(recon_attrs (B,18), attrs (B,18) float,
    size_average=False)`` (:56), which no torch version accepts for a float target; the attribute decoder ends in a sigmoid and
    the training loss is a Bernoulli cross entropy (celeba/train.py:68-72), so the summed binary cross entropy is what is
    computed here."""
    from .celeba import _BCEMeanFn as _BCE
    assert not (image_only and attrs_only)
    model.eval()
    test_image_nll, test_attrs_nll, n_seen = 0.0, 0.0, 0
    for image, attrs in loader:
        if use_cuda:
            image, attrs = image.cuda(), attrs.cuda()
        if not image_only and not attrs_only:
            _, _, mu, logvar = model(image, attrs)
        elif image_only:
            _, _, mu, logvar = model(image=image)
        else:
            _, _, mu, logvar = model(attrs=attrs)
        batch_size, n_latents = mu.size(0), mu.size(1)
        sample = torch.randn(n_samples, n_latents)                # drawn on the host like the reference (:40)
        if use_cuda:
            sample = sample.cuda()
        z = sample.unsqueeze(0) * logvar.mul(0.5).exp().unsqueeze(1) + mu.unsqueeze(1)      # (B, n_samples, D)
        image_nll, attrs_nll = 0.0, 0.0
        for i in range(n_samples):
            zi = z[:, i].contiguous()
            recon_image = model.image_decoder(zi)
            recon_attrs = model.attrs_decoder(zi)
            image_nll += float(_BCE.apply(recon_image.reshape(batch_size, -1), image.reshape(batch_size, -1))) * image.numel()
            attrs_nll += float(_BCE.apply(recon_attrs.reshape(batch_size, -1), attrs.reshape(batch_size, -1))) * attrs.numel()
        test_image_nll += image_nll / n_samples
        test_attrs_nll += attrs_nll / n_samples
        n_seen += batch_size
        if verbose:
            print('Evaluating: [{}/{}]'.format(n_seen, len(loader) * batch_size))
    return test_image_nll / max(n_seen, 1), test_attrs_nll / max(n_seen, 1)


@torch.no_grad()
def test_celeba(model, loader, use_cuda=True, verbose=True):
    """-> (joint, image-only, attribute-only) eval-mode losses averaged over the batches with kl_lambda = 1, celeba/test.py:44-71."""
    from .celeba import loss_function
    model.eval()
    sums, nb = [0.0, 0.0, 0.0], 0
    for image, attrs in loader:
        if use_cuda:
            image, attrs = image.cuda(), attrs.cuda()
        for k, kw in enumerate((dict(image=image, attrs=attrs), dict(image=image), dict(attrs=attrs))):
            recon_image, recon_attrs, mu, logvar = model(**kw)
            sums[k] += float(loss_function(mu, logvar, recon_x=recon_image, x=image, recon_y=recon_attrs, y=attrs,
                                           kl_lambda=1., lambda_x=1., lambda_y=1.))
        nb += 1
    out = tuple(s / max(nb, 1) for s in sums)
    if verbose:
        print('====> Test Epoch\tJoint loss: {:.4f}\tImage loss: {:.4f}\tAttrs loss:{:.4f}'.format(*out))
    return out


test_celeba.__test__ = False              # not a pytest test


# ------------------------------------------------------------------------------------------------------------------
# multimnist/sample.py:59-144 as a function and a script: `python -m multimodal_vae_amd.evaluate sample model.pth.tar ...`

@torch.no_grad()
def sample(vae, n_samples=64, image=None, text=None, use_cuda=True):
    """Unconditional (image = text = None) or conditional generation: (image samples (n,1,50,50), token samples (n,4) LongTensor).
    ``image``: (1,1,50,50) float in [0,1]; ``text``: (1,4) LongTensor (utils.char_tensor) -- multimnist/sample.py:80-130."""
    vae.eval()
    n_latents = vae.n_latents
    dev = torch.device("cuda") if use_cuda else torch.device("cpu")
    if image is None and text is None:
        mu, std = torch.zeros(1, device=dev), torch.ones(1, device=dev)
    elif text is None:
        mu, logvar = vae.encode_image(image.to(dev))
        std = logvar.mul(0.5).exp()
    elif image is None:
        mu, logvar = vae.encode_text(text.to(dev))
        std = logvar.mul(0.5).exp()
    else:
        image_mu, image_logvar = vae.encode_image(image.to(dev))
        text_mu, text_logvar = vae.encode_text(text.to(dev))
        mu, logvar = vae.experts(torch.stack((image_mu, text_mu), dim=0), torch.stack((image_logvar, text_logvar), dim=0))
        std = logvar.mul(0.5).exp()
    z = torch.randn(n_samples, n_latents).to(dev)
    z = z * std.expand_as(z) + mu.expand_as(z)
    image_recon = vae.decode_image(z.contiguous()).cpu().view(n_samples, 1, 50, 50)
    text_recon = torch.max(vae.decode_text(z.contiguous()).cpu(), dim=2)[1]
    return image_recon, text_recon


def _main(argv=None):
    import argparse
    import os
    from .train import load_checkpoint
    from .utils import char_tensor, tensor_to_string
    parser = argparse.ArgumentParser(prog="python -m multimodal_vae_amd.evaluate")
    sub = parser.add_subparsers(dest="cmd", required=True)
    ps = sub.add_parser("sample", help="multimnist/sample.py")
    ps.add_argument('model_path', type=str, help='path to trained model file.')
    ps.add_argument('--n_samples', type=int, default=64, help='Number of images and texts to sample.')
    ps.add_argument('--condition_on_image', type=str, default=None, help='a .pt file holding a (50,50) uint8 or float image')
    ps.add_argument('--condition_on_text', type=str, default=None, help='a digit string of at most 4 characters')
    ps.add_argument('--out', type=str, default='./results')
    args = parser.parse_args(argv)
    vae = load_checkpoint(args.model_path, use_cuda=True)
    image = text = None
    if args.condition_on_image:
        im = torch.load(args.condition_on_image)
        image = (im.float() / 255.0 if im.dtype == torch.uint8 else im.float()).view(1, 1, 50, 50)
    if args.condition_on_text:
        text = char_tensor(args.condition_on_text).unsqueeze(0)
    image_recon, text_recon = sample(vae, args.n_samples, image, text)
    os.makedirs(args.out, exist_ok=True)                          # (the reference calls the non-existent os.mkdirs, sample.py:133)
    torch.save(image_recon, os.path.join(args.out, 'sample_image.pt'))      # no torchvision here: the tensor, not a PNG grid
    with open(os.path.join(args.out, 'sample_text.txt'), 'w') as fp:
        for i in range(text_recon.size(0)):
            fp.write('%s\n' % tensor_to_string(text_recon[i]))


if __name__ == "__main__":
    _main()
