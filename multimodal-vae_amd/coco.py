"""Drop-in Python face of the reference's ``coco/model.py`` + ``loss_function`` (coco/train.py:66-84).

Same class names, constructor arguments, ``forward`` signatures (``vae(image=, text=)``) and ``state_dict`` keys as the
reference.  ``nn.*`` children are parameter containers only; every module forward/backward is a call into
libmmvae_hip.so.  No CPU fallback.  ``FusedTrainer`` = the train() closure body (coco/train.py:138-173) as one enqueue.

GloVe: the reference looks up exactly one vector at run time, ``GloVe('<s>')``, the decoder's first input
(coco/model.py:271-272; the ``</s>`` pre-fill of :275-277 is overwritten at every position).  It is passed in as ``sos``
(300 floats) and kept as a non-persistent buffer, so ``state_dict()`` has the reference's keys.  ``generate`` (nearest-word
decoding over the 2 GB GloVe table) is outside this engine; ``generate_vector`` is ``forward``.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
import torch.nn as nn

from ._lib import MMVAEError, call, ptr
from .core import CocoState, FusedCocoStep, StepOutputs
from .multimnist import (ProductOfExperts, Swish, _BCEMeanFn, _Core, _KLSumFn, _ModuleFn, _ReparamFn, _core_of,
                         _gscale, _seed_from_torch, _stream, swish)

MAX_WORDS = 102       # coco/utils.py:12-15
N_EMBEDDING = 300
N_HIDDENS = 200
DROP_P = 0.1


def _prep(mod: nn.Module, prefix: str, x: torch.Tensor):
    steps = mod.steps
    core = _core_of(mod, prefix, lambda n, d: CocoState(n, d, steps))
    st = core.sync(x.device)
    B = x.shape[0]
    names = [prefix + k for k, _ in mod.named_parameters()]
    plist = [p for _, p in mod.named_parameters()]
    return core, st, B, st.plan(B), st.module_workspace_bytes(B), names, plist


def _keep(shape, device, salt):
    m = torch.empty(*shape, dtype=torch.uint8, device=device)
    call("mmvae_keep_mask", ptr(m), m.numel(), DROP_P, _seed_from_torch(), None, salt, _stream())
    return m


class ImageEncoder(nn.Module):
    """coco/model.py:147-187"""

    def __init__(self, n_latents, steps=MAX_WORDS):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 4, 2, 1, bias=False), Swish(),
            nn.Conv2d(64, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.Conv2d(128, 256, 4, 2, 1, bias=False), nn.BatchNorm2d(256), Swish(),
            nn.Conv2d(256, 512, 4, 2, 1, bias=False), nn.BatchNorm2d(512), Swish())
        self.classifier = nn.Sequential(nn.Linear(512 * 2 * 2, 1024), Swish(), nn.Dropout(p=0.1),
                                        nn.Linear(1024, 256), Swish(), nn.Dropout(p=0.1), nn.Linear(256, n_latents * 2))
        self.n_latents = n_latents
        self.steps = steps
        self._core = None

    def forward(self, x, masks=None):
        n = self.n_latents
        x = x.contiguous().float()
        assert x.shape[1:] == (3, 32, 32), "expected (B,3,32,32) images (coco/train.py:107-112)"
        core, st, B, h, wsb, names, plist = _prep(self, "image_encoder.", x)
        p1, p2 = self.classifier[2].p, self.classifier[5].p
        m1 = m2 = None
        if self.training and (p1 > 0 or p2 > 0):
            if abs(p1 - DROP_P) > 1e-9 or abs(p2 - DROP_P) > 1e-9:
                raise MMVAEError("the HIP image encoder supports Dropout p in {0, 0.1} on both layers (reference: 0.1)")
            if masks is not None:
                m1, m2 = (m.to(torch.uint8).contiguous() for m in masks)
            else:
                m1, m2 = _keep((B, 1024), x.device, 2), _keep((B, 256), x.device, 3)
        training = int(self.training)

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            out = torch.empty(B, 2 * n, dtype=torch.float32, device=x.device)
            call("mmvae_coco_image_encoder_fwd", h, ptr(ws), wsb, ptr(x), ptr(m1), ptr(m2), training, ptr(out), _stream())
            ctx.ws = ws
            return out

        def bwd(ctx, d_out):
            st.grads.zero_()
            call("mmvae_coco_image_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_out.contiguous()), ptr(m1), ptr(m2), _stream())
            return [None] + core.grads_for(names)

        out = _ModuleFn.apply(fwd, bwd, 1, x, *plist)
        return out[:, :n], out[:, n:]


class ImageDecoder(nn.Module):
    """coco/model.py:190-216"""

    def __init__(self, n_latents, steps=MAX_WORDS):
        super().__init__()
        self.upsample = nn.Sequential(nn.Linear(n_latents, 512 * 2 * 2), Swish())
        self.hallucinate = nn.Sequential(
            nn.ConvTranspose2d(512, 256, 4, 2, 1, bias=False), nn.BatchNorm2d(256), Swish(),
            nn.ConvTranspose2d(256, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.ConvTranspose2d(64, 3, 4, 2, 1, bias=False))
        self.n_latents = n_latents
        self.steps = steps
        self._core = None

    def forward(self, z):
        z = z.contiguous().float()
        core, st, B, h, wsb, names, plist = _prep(self, "image_decoder.", z)
        training = int(self.training)
        n = self.n_latents

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
            recon = torch.empty(B, 3, 32, 32, dtype=torch.float32, device=z.device)
            call("mmvae_coco_image_decoder_fwd", h, ptr(ws), wsb, ptr(z), training, ptr(recon), _stream())
            ctx.ws, ctx.recon = ws, recon
            return recon

        def bwd(ctx, d_recon):
            st.grads.zero_()
            dz = torch.empty(B, n, dtype=torch.float32, device=z.device)
            call("mmvae_coco_image_decoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_recon.contiguous()), ptr(ctx.recon), ptr(dz), _stream())
            return [dz] + core.grads_for(names)

        return _ModuleFn.apply(fwd, bwd, 1, z, *plist)


class TextEncoder(nn.Module):
    """coco/model.py:219-245: biGRU(300 -> 200) over (B, steps, 300) GloVe vectors -> (mu, logvar)."""

    def __init__(self, n_latents, n_embedding=N_EMBEDDING, steps=MAX_WORDS):
        super().__init__()
        if n_embedding != N_EMBEDDING:
            raise MMVAEError("the HIP TextEncoder implements n_embedding=300 (GloVe-840B)")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # dropout on a 1-layer GRU: a no-op the reference also asks for
            self.gru = nn.GRU(n_embedding, N_HIDDENS, 1, dropout=0.1, bidirectional=True)
        self.h2p = nn.Linear(N_HIDDENS, n_latents * 2)
        self.n_latents = n_latents
        self.n_embedding = n_embedding
        self.steps = steps
        self._core = None

    def forward(self, x):
        n = self.n_latents
        x = x.contiguous().float()
        assert x.dim() == 3 and x.shape[1] == self.steps and x.shape[2] == N_EMBEDDING, \
            "expected (B, %d, 300) caption tensors (coco/utils.py:36-47)" % self.steps
        core, st, B, h, wsb, names, plist = _prep(self, "text_encoder.", x)

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            out = torch.empty(B, 2 * n, dtype=torch.float32, device=x.device)
            call("mmvae_coco_text_encoder_fwd", h, ptr(ws), wsb, ptr(x), ptr(out), _stream())
            ctx.ws = ws
            return out

        def bwd(ctx, d_out):
            st.grads.zero_()
            call("mmvae_coco_text_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(x), ptr(d_out.contiguous()), _stream())
            return [None] + core.grads_for(names)

        out = _ModuleFn.apply(fwd, bwd, 1, x, *plist)
        return out[:, :n], out[:, n:]


class TextDecoder(nn.Module):
    """coco/model.py:248-312: 2-layer GRU regressing one 300-d vector per step, fed back as the next input."""

    def __init__(self, n_latents, n_embedding=N_EMBEDDING, n_hiddens=N_HIDDENS, use_cuda=False, sos=None, steps=MAX_WORDS):
        super().__init__()
        if n_embedding != N_EMBEDDING or n_hiddens != N_HIDDENS:
            raise MMVAEError("the HIP TextDecoder implements n_embedding=300, n_hiddens=200")
        self.z2h = nn.Linear(n_latents, n_hiddens)
        self.gru = nn.GRU(n_embedding + n_latents, n_hiddens, 2, dropout=0.1)
        self.h2o = nn.Linear(n_hiddens + n_latents, n_embedding)
        if sos is None:
            raise MMVAEError("TextDecoder needs sos= the 300-d GloVe vector of '<s>' (coco/model.py:271); the GloVe table "
                             "itself is not part of this engine")
        self.register_buffer("sos", torch.as_tensor(sos, dtype=torch.float32).reshape(n_embedding).clone(), persistent=False)
        self.use_cuda = use_cuda
        self.n_latents = n_latents
        self.n_embedding = n_embedding
        self.n_hiddens = n_hiddens
        self.steps = steps
        self._core = None

    def forward(self, z, keep: Optional[torch.Tensor] = None):
        z = z.contiguous().float()
        core, st, B, h, wsb, names, plist = _prep(self, "text_decoder.", z)
        training = int(self.training)
        T, n = self.steps, self.n_latents
        pdrop = self.gru.dropout
        if self.training and pdrop > 0:
            if abs(pdrop - DROP_P) > 1e-9:
                raise MMVAEError("the HIP text decoder supports GRU dropout in {0, 0.1} (reference: 0.1)")
            keep = _keep((T, B, N_HIDDENS), z.device, 4) if keep is None else keep.to(torch.uint8).contiguous()
        else:
            keep = None
        sos = self.sos.to(z.device).contiguous()

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
            sentence = torch.empty(B, T, N_EMBEDDING, dtype=torch.float32, device=z.device)
            call("mmvae_coco_text_decoder_fwd", h, ptr(ws), wsb, ptr(z), ptr(sos), ptr(keep), training, ptr(sentence), _stream())
            ctx.ws, ctx.sentence = ws, sentence
            return sentence

        def bwd(ctx, d_sentence):
            st.grads.zero_()
            dz = torch.empty(B, n, dtype=torch.float32, device=z.device)
            call("mmvae_coco_text_decoder_bwd", h, ptr(ctx.ws), wsb, ptr(z), ptr(sos), ptr(keep), ptr(ctx.sentence),
                 ptr(d_sentence.contiguous()), ptr(dz), _stream())
            return [dz] + core.grads_for(names)

        return _ModuleFn.apply(fwd, bwd, 1, z, *plist)

    def generate_vector(self, z):
        """coco/model.py:308-309"""
        return self.forward(z)

    def generate(self, z):
        raise MMVAEError("TextDecoder.generate (coco/model.py:290-306) decodes vectors to words through the GloVe table, "
                         "which is not part of this engine; use generate_vector")


class MultimodalVAE(nn.Module):
    """coco/model.py:22-90"""

    def __init__(self, n_latents=20, use_cuda=False, sos=None, steps=MAX_WORDS):
        super().__init__()
        self.image_encoder = ImageEncoder(n_latents, steps=steps)
        self.image_decoder = ImageDecoder(n_latents, steps=steps)
        self.text_encoder = TextEncoder(n_latents, steps=steps)
        self.text_decoder = TextDecoder(n_latents, use_cuda=use_cuda, sos=sos, steps=steps)
        self.experts = ProductOfExperts()
        self.n_latents = n_latents
        self.steps = steps
        self._core = _Core(self, "", n_latents, lambda n, d: CocoState(n, d, steps))
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

    def decode_image(self, z):
        return self.image_decoder(z)

    def encode_text(self, x):
        return self.text_encoder(x)

    def decode_text(self, z):
        return self.text_decoder(z)

    def prior(self, size, use_cuda=False):
        """coco/model.py:50-58"""
        mu, logvar = torch.zeros(size), torch.log(torch.ones(size))
        if use_cuda:
            mu, logvar = mu.cuda(), logvar.cuda()
        return mu, logvar

    def forward(self, image=None, text=None, eps=None, enc_masks=None, gru_keep=None):
        assert image is not None or text is not None
        if image is not None and text is not None:
            image_mu, image_logvar = self.image_encoder(image, enc_masks)
            text_mu, text_logvar = self.text_encoder(text)
            mu = torch.stack((image_mu, text_mu), dim=0)
            logvar = torch.stack((image_logvar, text_logvar), dim=0)
        elif image is not None:
            mu, logvar = self.image_encoder(image, enc_masks)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        else:
            mu, logvar = self.text_encoder(text)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        mu, logvar = self.experts(mu, logvar)
        z = self.reparametrize(mu, logvar, eps)
        return self.image_decoder(z), self.text_decoder(z, gru_keep), mu, logvar



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
        call("mmvae_mse_bwd", ptr(a), ptr(b), a.numel(), 1.0 / a.numel(), ptr(_gscale(g)), ptr(da), _stream())
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


class FusedTrainer:
    """``FusedTrainer(vae, batch_size, lr)(image, text)`` == zero_grad + 3 passes + 3 losses + backward + Adam step
    (coco/train.py:138-173, lr default 1e-4 as coco/train.py:94) on ``vae``'s own parameters."""

    def __init__(self, vae: MultimodalVAE, batch_size: int, lr: float = 1e-4, kl_lambda: float = 1e-3, seed: int = 1234,
                 world_size: int = 1, all_reduce=None):
        dev = next(vae.parameters()).device
        self.vae = vae
        st = vae._core.sync(dev)
        self.engine = FusedCocoStep(st, batch_size, vae.text_decoder.sos, lr=lr, kl_lambda=kl_lambda, seed=seed,
                                    world_size=world_size, all_reduce=all_reduce)

    def __call__(self, image, text, **kw) -> StepOutputs:
        self.engine.enc_dropout = self.vae.image_encoder.classifier[2].p > 0
        self.engine.gru_dropout = self.vae.text_decoder.gru.dropout > 0
        return self.engine(image, text, **kw)

    def evaluate(self, image, text, **kw) -> StepOutputs:
        return self.engine.forward_backward(image, text, training=False, backward=False, **kw)
