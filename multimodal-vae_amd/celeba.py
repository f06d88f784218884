"""Drop-in Python face of the reference's ``celeba/model.py`` + ``loss_function`` (celeba/train.py:60-81).

Same class names, constructor arguments, ``forward`` signatures (``vae(image=, attrs=)``) and ``state_dict`` keys as
the reference.  ``nn.*`` children are parameter containers only; every module forward/backward is a call into
libmmvae_hip.so.  No CPU fallback.  ``FusedTrainer`` = the train() closure body (celeba/train.py:131-149) as one enqueue.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
import torch.nn as nn

from ._lib import MMVAEError, call, ptr
from .core import CelebaState, FusedCelebaStep, StepOutputs
from .multimnist import (ProductOfExperts, Swish, _BCEMeanFn, _Core, _KLSumFn, _ModuleFn, _ReparamFn, _core_of,
                         _seed_from_torch, _stream, swish)

N_ATTRS = 18          # celeba/datasets.py:26-28
DROP_P = 0.1


def _prep(mod: nn.Module, prefix: str, x: torch.Tensor):
    core = _core_of(mod, prefix, CelebaState)
    st = core.sync(x.device)
    B = x.shape[0]
    names = [prefix + k for k, _ in mod.named_parameters()]
    plist = [p for _, p in mod.named_parameters()]
    return core, st, B, st.plan(B), st.module_workspace_bytes(B), names, plist


class ImageEncoder(nn.Module):
    """celeba/model.py:91-128"""

    def __init__(self, n_latents):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 32, 4, 2, 1, bias=False), Swish(),
            nn.Conv2d(32, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.Conv2d(64, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.Conv2d(128, 256, 4, 1, 0, bias=False), nn.BatchNorm2d(256), Swish())
        self.classifier = nn.Sequential(nn.Linear(256 * 5 * 5, 1024), Swish(), nn.Dropout(p=0.1), nn.Linear(1024, n_latents * 2))
        self.n_latents = n_latents
        self._core = None

    def weight_init(self, mean, std):
        """celeba/model.py:121-123: iterates the two nn.Sequential containers, so it initialises nothing (kept as is)."""
        for m in self._modules:
            normal_init(self._modules[m], mean, std)

    def forward(self, x, mask: Optional[torch.Tensor] = None):
        n = self.n_latents
        x = x.contiguous().float()
        assert x.shape[1:] == (3, 64, 64), "expected (B,3,64,64) images"
        core, st, B, h, wsb, names, plist = _prep(self, "image_encoder.", x)
        p = self.classifier[2].p
        m1 = None
        if self.training and p > 0:
            if abs(p - DROP_P) > 1e-9:
                raise MMVAEError("the HIP image encoder supports Dropout p in {0, 0.1} (reference: 0.1)")
            if mask is not None:
                m1 = mask.to(torch.uint8).contiguous()
            else:
                m1 = torch.empty(B, 1024, dtype=torch.uint8, device=x.device)
                call("mmvae_keep_mask", ptr(m1), m1.numel(), DROP_P, _seed_from_torch(), None, 2, _stream())
        training = int(self.training)

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            out = torch.empty(B, 2 * n, dtype=torch.float32, device=x.device)
            call("mmvae_celeba_image_encoder_fwd", h, ptr(ws), wsb, ptr(x), ptr(m1), training, ptr(out), _stream())
            ctx.ws = ws
            return out

        def bwd(ctx, d_out):
            st.grads.zero_()
            call("mmvae_celeba_image_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_out.contiguous()), ptr(m1), _stream())
            return [None] + core.grads_for(names)

        out = _ModuleFn.apply(fwd, bwd, 1, x, *plist)
        return out[:, :n], out[:, n:]


class ImageDecoder(nn.Module):
    """celeba/model.py:131-161"""

    def __init__(self, n_latents):
        super().__init__()
        self.upsample = nn.Sequential(nn.Linear(n_latents, 256 * 5 * 5), Swish())
        self.hallucinate = nn.Sequential(
            nn.ConvTranspose2d(256, 128, 4, 1, 0, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.ConvTranspose2d(64, 32, 4, 2, 1, bias=False), nn.BatchNorm2d(32), Swish(),
            nn.ConvTranspose2d(32, 3, 4, 2, 1, bias=False))
        self.n_latents = n_latents
        self._core = None

    def weight_init(self, mean, std):
        for m in self._modules:
            normal_init(self._modules[m], mean, std)

    def forward(self, z):
        z = z.contiguous().float()
        core, st, B, h, wsb, names, plist = _prep(self, "image_decoder.", z)
        training = int(self.training)
        n = self.n_latents

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
            recon = torch.empty(B, 3, 64, 64, dtype=torch.float32, device=z.device)
            call("mmvae_celeba_image_decoder_fwd", h, ptr(ws), wsb, ptr(z), training, ptr(recon), _stream())
            ctx.ws, ctx.recon = ws, recon
            return recon

        def bwd(ctx, d_recon):
            st.grads.zero_()
            dz = torch.empty(B, n, dtype=torch.float32, device=z.device)
            call("mmvae_celeba_image_decoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_recon.contiguous()), ptr(ctx.recon), ptr(dz), _stream())
            return [dz] + core.grads_for(names)

        return _ModuleFn.apply(fwd, bwd, 1, z, *plist)


class AttributeEncoder(nn.Module):
    """celeba/model.py:164-178"""

    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(N_ATTRS, 64), nn.BatchNorm1d(64), Swish(), nn.Linear(64, n_latents * 2))
        self.n_latents = n_latents
        self._core = None

    def forward(self, x):
        n = self.n_latents
        x = x.contiguous().float()
        assert x.dim() == 2 and x.shape[1] == N_ATTRS
        core, st, B, h, wsb, names, plist = _prep(self, "attrs_encoder.", x)
        if self.training and B <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        training = int(self.training)

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            out = torch.empty(B, 2 * n, dtype=torch.float32, device=x.device)
            call("mmvae_celeba_attrs_encoder_fwd", h, ptr(ws), wsb, ptr(x), training, ptr(out), _stream())
            ctx.ws = ws
            return out

        def bwd(ctx, d_out):
            st.grads.zero_()
            call("mmvae_celeba_attrs_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_out.contiguous()), _stream())
            return [None] + core.grads_for(names)

        out = _ModuleFn.apply(fwd, bwd, 1, x, *plist)
        return out[:, :n], out[:, n:]


class AttributeDecoder(nn.Module):
    """celeba/model.py:181-196"""

    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(n_latents, 64), nn.BatchNorm1d(64), Swish(), nn.Linear(64, N_ATTRS))
        self.n_latents = n_latents
        self._core = None

    def forward(self, z):
        z = z.contiguous().float()
        core, st, B, h, wsb, names, plist = _prep(self, "attrs_decoder.", z)
        if self.training and B <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        training = int(self.training)
        n = self.n_latents

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
            recon = torch.empty(B, N_ATTRS, dtype=torch.float32, device=z.device)
            call("mmvae_celeba_attrs_decoder_fwd", h, ptr(ws), wsb, ptr(z), training, ptr(recon), _stream())
            ctx.ws, ctx.recon = ws, recon
            return recon

        def bwd(ctx, d_recon):
            st.grads.zero_()
            dz = torch.empty(B, n, dtype=torch.float32, device=z.device)
            call("mmvae_celeba_attrs_decoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_recon.contiguous()), ptr(ctx.recon), ptr(dz), _stream())
            return [dz] + core.grads_for(names)

        return _ModuleFn.apply(fwd, bwd, 1, z, *plist)


def normal_init(m, mean, std):
    """celeba/model.py:214-217"""
    if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
        m.weight.data.normal_(mean, std)
        m.bias.data.zero_()


class MultimodalVAE(nn.Module):
    """celeba/model.py:14-57"""

    def __init__(self, n_latents=20, use_cuda=False):
        super().__init__()
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.attrs_encoder = AttributeEncoder(n_latents)
        self.attrs_decoder = AttributeDecoder(n_latents)
        self.experts = ProductOfExperts()
        self.n_latents = n_latents
        self._core = _Core(self, "", n_latents, CelebaState)
        for m in (self.image_encoder, self.image_decoder, self.attrs_encoder, self.attrs_decoder):
            object.__setattr__(m, "_mmvae_root", weakref.ref(self))

    def weight_init(self, mean, std):
        self.image_encoder.weight_init(mean=mean, std=std)
        self.image_decoder.weight_init(mean=mean, std=std)

    def reparametrize(self, mu, logvar, eps: Optional[torch.Tensor] = None):
        if self.training:
            if eps is None:
                eps = torch.empty_like(mu)
                call("mmvae_normal", ptr(eps), eps.numel(), _seed_from_torch(), None, 1, _stream())
            return _ReparamFn.apply(mu, logvar, eps.contiguous())
        return mu

    def forward(self, image=None, attrs=None, eps=None, enc_mask=None):
        assert image is not None or attrs is not None
        if image is not None and attrs is not None:
            image_mu, image_logvar = self.image_encoder(image, enc_mask)
            attrs_mu, attrs_logvar = self.attrs_encoder(attrs)
            mu = torch.stack((image_mu, attrs_mu), dim=0)
            logvar = torch.stack((image_logvar, attrs_logvar), dim=0)
        elif image is not None:
            mu, logvar = self.image_encoder(image, enc_mask)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        else:
            mu, logvar = self.attrs_encoder(attrs)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        mu, logvar = self.experts(mu, logvar)
        z = self.reparametrize(mu, logvar, eps)
        return self.image_decoder(z), self.attrs_decoder(z), mu, logvar


def loss_function(mu, logvar, recon_x=None, x=None, recon_y=None, y=None, kl_lambda=1e-3, lambda_x=1., lambda_y=1.):
    """celeba/train.py:60-81.  The per-attribute BCE loop averaged over attributes equals the mean over all B*18
    elements, which is what the kernel sums."""
    batch_size = mu.size(0)
    x_BCE, y_BCE = 0, 0
    if recon_x is not None and x is not None:
        x_BCE = _BCEMeanFn.apply(recon_x.reshape(-1, 3 * 64 * 64), x.reshape(-1, 3 * 64 * 64))
    if recon_y is not None and y is not None:
        y_BCE = _BCEMeanFn.apply(recon_y.reshape(-1, N_ATTRS), y.reshape(-1, N_ATTRS))
    KLD = _KLSumFn.apply(mu, logvar)
    KLD = KLD / batch_size * kl_lambda
    return lambda_x * x_BCE + lambda_y * y_BCE + KLD


elbo_loss = loss_function


class FusedTrainer:
    """``FusedTrainer(vae, batch_size, lr)(image, attrs)`` == zero_grad + 3 passes + 3 losses + backward + Adam step
    (celeba/train.py:131-149) on ``vae``'s own parameters."""

    def __init__(self, vae: MultimodalVAE, batch_size: int, lr: float = 1e-3, kl_lambda: float = 1e-3, seed: int = 1234,
                 world_size: int = 1, all_reduce=None):
        dev = next(vae.parameters()).device
        self.vae = vae
        st = vae._core.sync(dev)
        self.engine = FusedCelebaStep(st, batch_size, lr=lr, kl_lambda=kl_lambda, seed=seed, world_size=world_size,
                                      all_reduce=all_reduce)

    def __call__(self, image, attrs, **kw) -> StepOutputs:
        self.engine.enc_dropout = self.vae.image_encoder.classifier[2].p > 0
        return self.engine(image, attrs, **kw)

    def evaluate(self, image, attrs, **kw) -> StepOutputs:
        return self.engine.forward_backward(image, attrs, training=False, backward=False, **kw)
