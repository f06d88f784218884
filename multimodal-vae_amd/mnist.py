"""Drop-in Python face of the reference's ``mnist/model.py`` + ``loss_function`` (mnist/train.py:64-81).

Same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys as the reference.  The
``nn.Linear / nn.BatchNorm1d / nn.Embedding`` children are parameter containers only; every module forward/backward
is a call into libmmvae_hip.so.  No CPU fallback.  ``FusedTrainer`` runs the whole train() closure body
(mnist/train.py:131-147 + optimizer.step()) as one enqueue.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
import torch.nn as nn

from ._lib import MMVAEError, call, ptr
from .core import FusedMnistStep, MnistState, StepOutputs
from .multimnist import (ProductOfExperts, _BCEMeanFn, _Core, _KLSumFn, _ModuleFn, _NLLMeanFn, _ReparamFn, _core_of,
                         _seed_from_torch, _stream)


def _run_module(mod: nn.Module, prefix: str, x: torch.Tensor, out_shape, name: str, needs_input_grad: bool, extra_fwd=(),
                bwd_args=None):
    """Shared bridge of the four MLP modules: forward = mmvae_mnist_<name>_fwd, backward = ..._bwd."""
    core = _core_of(mod, prefix, MnistState)
    st = core.sync(x.device)
    B = x.shape[0]
    if mod.training and B <= 1:
        raise ValueError("Expected more than 1 value per channel when training")
    h = st.plan(B)
    wsb = st.module_workspace_bytes(B)
    training = int(mod.training)
    names = [prefix + k for k, _ in mod.named_parameters()]
    plist = [p for _, p in mod.named_parameters()]
    n_lat = core.n_latents

    def fwd(ctx):
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        out = torch.empty((B,) + tuple(out_shape), dtype=torch.float32, device=x.device)
        call("mmvae_mnist_%s_fwd" % name, h, ptr(ws), wsb, ptr(x), training, ptr(out), _stream())
        ctx.ws, ctx.out = ws, out
        return out

    def bwd(ctx, d_out):
        st.grads.zero_()
        d = d_out.contiguous()
        if name == "image_encoder":
            call("mmvae_mnist_image_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(d), _stream())
            dx = None
        elif name == "text_encoder":
            call("mmvae_mnist_text_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(x), ptr(d), _stream())
            dx = None
        else:
            dx = torch.empty(B, n_lat, dtype=torch.float32, device=x.device)
            call("mmvae_mnist_%s_bwd" % name, h, ptr(ctx.ws), wsb, ptr(d), ptr(ctx.out), ptr(dx), _stream())
        return [dx] + core.grads_for(names)

    return _ModuleFn.apply(fwd, bwd, 1, x, *plist)


class ImageEncoder(nn.Module):
    """mnist/model.py:99-118"""

    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(784, 400), nn.BatchNorm1d(400), nn.ReLU(),
            nn.Linear(400, 200), nn.BatchNorm1d(200), nn.ReLU(),
            nn.Linear(200, n_latents * 2))
        self.n_latents = n_latents
        self._core = None

    def forward(self, x):
        n = self.n_latents
        x = x.contiguous().float()
        assert x.dim() == 2 and x.shape[1] == 784, "expected (B,784) flattened images (mnist/train.py:123)"
        out = _run_module(self, "image_encoder.", x, (2 * n,), "image_encoder", False)
        return out[:, :n], out[:, n:]


class ImageDecoder(nn.Module):
    """mnist/model.py:121-133"""

    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(n_latents, 200), nn.BatchNorm1d(200), nn.ReLU(),
            nn.Linear(200, 400), nn.BatchNorm1d(400), nn.ReLU(),
            nn.Linear(400, 784))
        self.n_latents = n_latents
        self._core = None

    def forward(self, z):
        return _run_module(self, "image_decoder.", z.contiguous().float(), (784,), "image_decoder", True)


class TextEncoder(nn.Module):
    """mnist/model.py:136-153"""

    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(nn.Embedding(10, 50), nn.BatchNorm1d(50), nn.ReLU(), nn.Linear(50, n_latents * 2))
        self.n_latents = n_latents
        self._core = None

    def forward(self, x):
        n = self.n_latents
        x = x.contiguous().long()
        assert x.dim() == 1, "expected (B,) digit labels"
        out = _run_module(self, "text_encoder.", x, (2 * n,), "text_encoder", False)
        return out[:, :n], out[:, n:]


class TextDecoder(nn.Module):
    """mnist/model.py:156-170"""

    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(n_latents, 10), nn.BatchNorm1d(10), nn.ReLU(), nn.Linear(10, 10))
        self.n_latents = n_latents
        self._core = None

    def forward(self, z):
        return _run_module(self, "text_decoder.", z.contiguous().float(), (10,), "text_decoder", True)


class MultimodalVAE(nn.Module):
    """mnist/model.py:14-96"""

    def __init__(self, n_latents=20, precision="fp32"):
        """``precision``: "fp32" (default; the reference's own arithmetic, fp32 MFMA) or "bf16" (bf16 MFMA operands)."""
        super().__init__()
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.text_encoder = TextEncoder(n_latents)
        self.text_decoder = TextDecoder(n_latents)
        self.experts = ProductOfExperts()
        self.n_latents = n_latents
        self.precision = precision
        self._core = _Core(self, "", n_latents, lambda n, d: MnistState(n, d, precision))
        for m in (self.image_encoder, self.image_decoder, self.text_encoder, self.text_decoder):
            object.__setattr__(m, "_mmvae_root", weakref.ref(self))

    def reparametrize(self, mu, logvar, eps: Optional[torch.Tensor] = None):
        if self.training:
            if eps is None:
                eps = torch.empty_like(mu)
                call("mmvae_normal", ptr(eps), eps.numel(), _seed_from_torch(), None, 1, _stream())
            return _ReparamFn.apply(mu, logvar, eps.contiguous())
        return mu

    def encode_image(self, x):
        return self.image_encoder(x)

    def decode_image(self, x):
        return self.image_decoder(x)

    def encode_text(self, x):
        return self.text_encoder(x)

    def decode_text(self, x):
        return self.text_decoder(x)

    def prior(self, size, use_cuda=False):
        mu = torch.zeros(size)
        logvar = torch.log(torch.ones(size))
        if use_cuda:
            mu, logvar = mu.cuda(), logvar.cuda()
        return mu, logvar

    def _latents(self, image, text):
        if image is not None and text is not None:
            image_mu, image_logvar = self.encode_image(image)
            text_mu, text_logvar = self.encode_text(text)
            mu = torch.stack((image_mu, text_mu), dim=0)
            logvar = torch.stack((image_logvar, text_logvar), dim=0)
        elif image is not None:
            mu, logvar = self.encode_image(image)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        else:
            mu, logvar = self.encode_text(text)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        return self.experts(mu, logvar)

    def forward(self, image=None, text=None, eps=None):
        assert image is not None or text is not None
        mu, logvar = self._latents(image, text)
        z = self.reparametrize(mu, logvar, eps)
        return self.decode_image(z), self.decode_text(z), mu, logvar

    def gen_latents(self, image, text):
        """mnist/model.py:86-96"""
        mu, logvar = self._latents(image, text)
        return self.reparametrize(mu, logvar)


def loss_function(mu, logvar, recon_image=None, image=None, recon_text=None, text=None,
                  kl_lambda=1e-3, lambda_xy=1., lambda_yx=1.):
    """mnist/train.py:64-81 (``kl_lambda`` is accepted and ignored, as in the reference)."""
    batch_size = mu.size(0)
    image_BCE, text_BCE = 0, 0
    if recon_image is not None and image is not None:
        image_BCE = lambda_xy * _BCEMeanFn.apply(recon_image.reshape(-1, 784), image.reshape(-1, 784))
    if recon_text is not None and text is not None:
        text_BCE = lambda_yx * _NLLMeanFn.apply(recon_text, text)
    KLD = _KLSumFn.apply(mu, logvar)
    KLD = KLD / (batch_size * (784 / 3))
    return image_BCE + text_BCE + KLD


elbo_loss = loss_function


class FusedTrainer:
    """``FusedTrainer(vae, batch_size, lr)(image, label)`` == zero_grad + 3 passes + 3 losses + backward + Adam step
    (mnist/train.py:131-147,149) on ``vae``'s own parameters."""

    def __init__(self, vae: MultimodalVAE, batch_size: int, lr: float = 1e-3, seed: int = 1234, world_size: int = 1,
                 all_reduce=None):
        dev = next(vae.parameters()).device
        self.vae = vae
        st = vae._core.sync(dev)
        self.engine = FusedMnistStep(st, batch_size, lr=lr, seed=seed, world_size=world_size, all_reduce=all_reduce)

    def __call__(self, image, label, **kw) -> StepOutputs:
        return self.engine(image.reshape(-1, 784), label, **kw)

    def evaluate(self, image, label, **kw) -> StepOutputs:
        """mnist/train.py:165-177: eval-mode forward of the 3 passes, no backward."""
        return self.engine.forward_backward(image.reshape(-1, 784), label, training=False, backward=False, **kw)
