"""Drop-in Python face of the reference's ``multimnist/model.py`` + ``loss_function`` (multimnist/train.py:69-87).

Same class names, constructor arguments, ``forward`` signatures, ``state_dict`` keys and train/eval behaviour as the
reference, so a reference-style ``train.py`` loop runs unchanged:

    vae = MultimodalVAE(n_latents=100, use_cuda=True).cuda()
    optimizer = optim.Adam(vae.parameters(), lr=1e-3)
    recon_image, recon_text, mu, logvar = vae(image, text)
    loss = loss_function(mu, logvar, recon_image=..., image=..., recon_text=..., text=..., kl_lambda=..., ...)
    loss.backward(); optimizer.step()

The ``nn.Conv2d / nn.Linear / nn.GRU ...`` children are kept ONLY as parameter containers (default PyTorch
initialisation, ``state_dict`` names): their ``forward`` is never called.  Every module forward/backward is a call
into libmmvae_hip.so (hand-written HIP kernels); there is no PyTorch/CPU fallback -- on a machine without a gfx950
GPU the first forward raises ``MMVAEError``.  The whole 3-pass training step as ONE fused enqueue is
``FusedTrainer`` below (what ``bench.py`` measures).
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import MMVAEError, call, ptr
from .core import FusedELBOStep, MultimnistState, StepOutputs

# multimnist/utils.py:14-19
max_length = 4
all_characters = '0123456789'
n_characters = len(all_characters)
SOS = n_characters
FILL = n_characters + 1
n_characters += 2

DROP_P = 0.1


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _seed_from_torch() -> int:
    """A 63-bit seed drawn from torch's default CPU generator (so torch.manual_seed makes runs reproducible)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


# ----------------------------------------------------------------------------------------------------------------
# shared device state of a module tree
# ----------------------------------------------------------------------------------------------------------------
class _Core:
    """Binds the nn.Parameters / BatchNorm buffers of a module tree to one flat ``MultimnistState``."""

    def __init__(self, owner: nn.Module, prefix: str, n_latents: int, state_cls=MultimnistState):
        self.state_cls = state_cls
        self.owner = weakref.ref(owner)
        self.prefix = prefix            # "" for MultimodalVAE, "image_encoder." for a standalone ImageEncoder, ...
        self.n_latents = n_latents
        self.state: Optional[MultimnistState] = None
        self._sig = None

    def _named(self):
        owner = self.owner()
        params = {self.prefix + n: p for n, p in owner.named_parameters()}
        bufs = {self.prefix + n: b for n, b in owner.named_buffers()}
        return owner, params, bufs

    def sync(self, device: torch.device) -> MultimnistState:
        """Make every parameter / BN buffer of the tree a view of the flat device buffers (no copy if it already is)."""
        if device.type != "cuda":
            raise MMVAEError("MMVAE HIP modules need CUDA/HIP tensors (got %s); move the module and its inputs to the GPU. "
                             "There is no CPU fallback." % device)
        if self.state is None or self.state.device != device:
            self.state = self.state_cls(self.n_latents, device)
            self._sig = None
        st = self.state
        st.ensure_packed()                 # a fused step may have deferred the refresh of the bf16 weight copies
        owner, params, bufs = self._named()
        base = st.params.data_ptr()
        for name, shape, off in st.table:
            p = params.get(name)
            if p is None:
                continue
            if p.device != device:
                raise MMVAEError("parameter %s is on %s but the input is on %s" % (name, p.device, device))
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                view = st.view(name)
                view.copy_(p.data.to(torch.float32).reshape(shape))
                p.data = view
                self._sig = None
        bbase = st.bn_stats.data_ptr()
        for i, (pre, c, off) in enumerate(st.bn_table):
            for j, nm in enumerate(("running_mean", "running_var")):
                b = bufs.get("%s.%s" % (pre, nm))
                if b is None:
                    continue
                if b.data_ptr() != bbase + 4 * (off + j * c):
                    view = st.bn_stats[off + j * c: off + (j + 1) * c]
                    view.copy_(b.to(device))
                    _set_buffer(owner, (pre + "." + nm)[len(self.prefix):], view)
            nb = bufs.get(pre + ".num_batches_tracked")
            if nb is not None and nb.data_ptr() != st.bn_nbt.data_ptr() + 8 * i:
                view = st.bn_nbt[i]
                view.copy_(nb.to(device))
                _set_buffer(owner, (pre + ".num_batches_tracked")[len(self.prefix):], view)
        # repack the bf16 GEMM copies when any parameter changed (optimizer.step(), load_state_dict, ...)
        sig = tuple(p._version for p in params.values())
        if sig != self._sig:
            st.pack_weights()
            self._sig = sig
        return st

    def param_list(self) -> List[nn.Parameter]:
        owner, params, _ = self._named()
        return [params[n] for n, _, _ in self.state.table if n in params]

    def grads_for(self, names: List[str]) -> List[torch.Tensor]:
        st = self.state
        out = []
        for n, shape, off in st.table:
            if n in names:
                numel = 1
                for s in shape:
                    numel *= s
                out.append(st.grads[off:off + numel].view(shape).clone())
        return out


def _set_buffer(root: nn.Module, dotted: str, value: torch.Tensor) -> None:
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        mod = getattr(mod, p)
    mod._buffers[parts[-1]] = value


def _core_of(module: nn.Module, prefix: str, state_cls=MultimnistState) -> _Core:
    root = getattr(module, "_mmvae_root", None)
    root = root() if root is not None else None
    if root is not None:
        return root._core
    if getattr(module, "_core", None) is None:
        module._core = _Core(module, prefix, module.n_latents, state_cls)
    return module._core


class _ModuleFn(torch.autograd.Function):
    """Generic bridge: forward/backward closures call the C-ABI; parameters are listed as inputs so autograd
    routes their gradients."""

    @staticmethod
    def forward(ctx, fwd, bwd, n_in, *tensors):
        ctx.bwd = bwd
        ctx.n_in = n_in
        out = fwd(ctx)
        return out

    @staticmethod
    def backward(ctx, *grads):
        res = ctx.bwd(ctx, *grads)
        return (None, None, None) + tuple(res)


# ----------------------------------------------------------------------------------------------------------------
# modules (parameter containers mirror multimnist/model.py exactly)
# ----------------------------------------------------------------------------------------------------------------
def swish(x):
    """multimnist/model.py:375-376"""
    return _Elementwise.apply(x)


class _Elementwise(torch.autograd.Function):
    # x*sigmoid(x) as a standalone op is not on the hot path (it is fused into the GEMM kernels); it exists only so
    # that ``Swish()(tensor)`` keeps working for user code.
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x * torch.sigmoid(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        s = torch.sigmoid(x)
        return g * s * (1 + x * (1 - s))


class Swish(nn.Module):
    """multimnist/model.py:379-381"""
    def forward(self, x):
        return swish(x)


class ImageEncoder(nn.Module):
    """multimnist/model.py:150-188"""

    def __init__(self, n_latents):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(1, 32, 4, 2, 1, bias=False), Swish(),
            nn.Conv2d(32, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.Conv2d(64, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.Conv2d(128, 256, 4, 2, 0, bias=False), nn.BatchNorm2d(256), Swish())
        self.classifier = nn.Sequential(
            nn.Linear(256 * 2 * 2, 400), Swish(), nn.Dropout(p=0.1),
            nn.Linear(400, 200), Swish(), nn.Dropout(p=0.1),
            nn.Linear(200, n_latents * 2))
        self.n_latents = n_latents
        self._core = None

    def forward(self, x, masks: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        n = self.n_latents
        core = _core_of(self, "image_encoder.")
        st = core.sync(x.device)
        x = x.contiguous().float()
        B = x.shape[0]
        assert x.shape[1:] == (1, 50, 50), "expected (B,1,50,50) images"
        if self.training and B * 625 <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        h = st.plan(B)
        wsb = st.module_workspace_bytes(B)
        p1, p2 = self.classifier[2].p, self.classifier[5].p
        m1 = m2 = None
        if self.training and (p1 > 0 or p2 > 0):
            if abs(p1 - DROP_P) > 1e-9 or abs(p2 - DROP_P) > 1e-9:
                raise MMVAEError("the HIP image encoder supports Dropout p in {0, 0.1} (reference: 0.1)")
            if masks is not None:
                m1, m2 = (m.to(torch.uint8).contiguous() for m in masks)
            else:
                m1 = torch.empty(B, 400, dtype=torch.uint8, device=x.device)
                m2 = torch.empty(B, 200, dtype=torch.uint8, device=x.device)
                seed = _seed_from_torch()
                call("mmvae_keep_mask", ptr(m1), m1.numel(), DROP_P, seed, None, 2, _stream())
                call("mmvae_keep_mask", ptr(m2), m2.numel(), DROP_P, seed, None, 3, _stream())
        training = int(self.training)
        names = ["image_encoder." + k for k, _ in self.named_parameters()]
        plist = [p for _, p in self.named_parameters()]

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            out = torch.empty(B, 2 * n, dtype=torch.float32, device=x.device)
            call("mmvae_mm_image_encoder_fwd", h, ptr(ws), wsb, ptr(x), ptr(m1), ptr(m2), training, ptr(out), _stream())
            ctx.ws = ws
            return out

        def bwd(ctx, d_out):
            st.grads.zero_()
            d = d_out.contiguous()
            call("mmvae_mm_image_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(d), ptr(m1), ptr(m2), _stream())
            return [None] + core.grads_for(names)          # no gradient w.r.t. the image

        out = _ModuleFn.apply(fwd, bwd, 1, x, *plist)
        return out[:, :n], out[:, n:]


class ImageDecoder(nn.Module):
    """multimnist/model.py:191-216"""

    def __init__(self, n_latents):
        super().__init__()
        self.upsample = nn.Sequential(nn.Linear(n_latents, 256 * 2 * 2), Swish())
        self.hallucinate = nn.Sequential(
            nn.ConvTranspose2d(256, 128, 4, 2, 0, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.ConvTranspose2d(64, 32, 5, 2, 1, bias=False), nn.BatchNorm2d(32), Swish(),
            nn.ConvTranspose2d(32, 1, 4, 2, 1, bias=False))
        self.n_latents = n_latents
        self._core = None

    def forward(self, z):
        core = _core_of(self, "image_decoder.")
        st = core.sync(z.device)
        z = z.contiguous().float()
        B = z.shape[0]
        if self.training and B * 36 <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        h = st.plan(B)
        wsb = st.module_workspace_bytes(B)
        training = int(self.training)
        names = ["image_decoder." + k for k, _ in self.named_parameters()]
        plist = [p for _, p in self.named_parameters()]

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
            recon = torch.empty(B, 1, 50, 50, dtype=torch.float32, device=z.device)
            call("mmvae_mm_image_decoder_fwd", h, ptr(ws), wsb, ptr(z), training, ptr(recon), _stream())
            ctx.ws, ctx.recon = ws, recon
            return recon

        def bwd(ctx, d_recon):
            st.grads.zero_()
            dz = torch.empty(B, self.n_latents, dtype=torch.float32, device=z.device)
            call("mmvae_mm_image_decoder_bwd", h, ptr(ctx.ws), wsb, ptr(d_recon.contiguous()), ptr(ctx.recon), ptr(dz), _stream())
            return [dz] + core.grads_for(names)

        return _ModuleFn.apply(fwd, bwd, 1, z, *plist)


class TextEncoder(nn.Module):
    """multimnist/model.py:219-247 (bidirectional=True is the only configuration MultimodalVAE uses)."""

    def __init__(self, n_latents, n_characters, n_hiddens=50, bidirectional=True):
        super().__init__()
        if n_hiddens != 100 or not bidirectional or n_characters != 12:
            raise MMVAEError("the HIP TextEncoder implements the configuration used by MultimodalVAE: "
                             "n_characters=12, n_hiddens=100, bidirectional=True")
        self.embed = nn.Embedding(n_characters, n_hiddens)
        self.gru = nn.GRU(n_hiddens, n_hiddens, 1, dropout=0.0, bidirectional=bidirectional)   # dropout is a no-op for 1 layer
        self.h2p = nn.Linear(n_hiddens, n_latents * 2)
        self.n_latents = n_latents
        self.n_hiddens = n_hiddens
        self.bidirectional = bidirectional
        self._core = None

    def forward(self, x):
        n = self.n_latents
        core = _core_of(self, "text_encoder.")
        st = core.sync(x.device)
        x = x.contiguous().long()
        B = x.shape[0]
        assert x.shape == (B, max_length)
        h = st.plan(B)
        wsb = st.module_workspace_bytes(B)
        names = ["text_encoder." + k for k, _ in self.named_parameters()]
        plist = [p for _, p in self.named_parameters()]

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            out = torch.empty(B, 2 * n, dtype=torch.float32, device=x.device)
            call("mmvae_mm_text_encoder_fwd", h, ptr(ws), wsb, ptr(x), ptr(out), _stream())
            ctx.ws = ws
            return out

        def bwd(ctx, d_out):
            st.grads.zero_()
            call("mmvae_mm_text_encoder_bwd", h, ptr(ctx.ws), wsb, ptr(x), ptr(d_out.contiguous()), _stream())
            return core.grads_for(names)

        out = _ModuleFn.apply(fwd, bwd, 0, *plist)
        return out[:, :n], out[:, n:]


class TextDecoder(nn.Module):
    """multimnist/model.py:250-307: 2-layer GRU, 4 greedy steps, log-softmax outputs (B, 4, 12)."""

    def __init__(self, n_latents, n_characters, n_hiddens=50, use_cuda=False):
        super().__init__()
        if n_hiddens != 100 or n_characters != 12:
            raise MMVAEError("the HIP TextDecoder implements the configuration used by MultimodalVAE: "
                             "n_characters=12, n_hiddens=100")
        self.embed = nn.Embedding(n_characters, n_hiddens)
        self.z2h = nn.Linear(n_latents, n_hiddens)
        self.gru = nn.GRU(n_hiddens + n_latents, n_hiddens, 2, dropout=0.1)
        self.h2o = nn.Linear(n_hiddens + n_latents, n_characters)
        self.use_cuda = use_cuda
        self.n_latents = n_latents
        self.n_characters = n_characters
        self._core = None
        self.last_tokens = None

    def forward(self, z, keep: Optional[torch.Tensor] = None, force_tokens: Optional[torch.Tensor] = None):
        core = _core_of(self, "text_decoder.")
        st = core.sync(z.device)
        z = z.contiguous().float()
        B = z.shape[0]
        h = st.plan(B)
        wsb = st.module_workspace_bytes(B)
        training = int(self.training)
        pdrop = self.gru.dropout
        if self.training and pdrop > 0:
            if abs(pdrop - DROP_P) > 1e-9:
                raise MMVAEError("the HIP text decoder supports GRU dropout in {0, 0.1} (reference: 0.1)")
            if keep is None:
                keep = torch.empty(max_length, B, 100, dtype=torch.uint8, device=z.device)
                call("mmvae_keep_mask", ptr(keep), keep.numel(), DROP_P, _seed_from_torch(), None, 4, _stream())
            else:
                keep = keep.to(torch.uint8).contiguous()
        else:
            keep = None
        ft = None if force_tokens is None else force_tokens.contiguous().long()
        names = ["text_decoder." + k for k, _ in self.named_parameters()]
        plist = [p for _, p in self.named_parameters()]

        def fwd(ctx):
            ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
            words = torch.empty(B, max_length, n_characters, dtype=torch.float32, device=z.device)
            tokens = torch.empty(B, max_length, dtype=torch.int64, device=z.device)
            call("mmvae_mm_text_decoder_fwd", h, ptr(ws), wsb, ptr(z), training, ptr(keep), ptr(ft), ptr(words), ptr(tokens), _stream())
            ctx.ws, ctx.words, ctx.tokens = ws, words, tokens
            self.last_tokens = tokens
            return words

        def bwd(ctx, d_words):
            st.grads.zero_()
            dz = torch.empty(B, self.n_latents, dtype=torch.float32, device=z.device)
            call("mmvae_mm_text_decoder_bwd", h, ptr(ctx.ws), wsb, ptr(z), ptr(keep), ptr(ft), ptr(ctx.words), ptr(ctx.tokens),
                 ptr(d_words.contiguous()), ptr(dz), _stream())
            return [dz] + core.grads_for(names)

        return _ModuleFn.apply(fwd, bwd, 1, z, *plist)

    def generate(self, z):
        """multimnist/model.py:290-296.  NB the reference samples ``torch.multinomial`` from LOG-probabilities (weights
        <= 0): under a modern torch that raises ("probability tensor contains ... element < 0"), and so does this method.
        The sampler runs on a host copy of the (B*4, 12) log-probabilities: on the GPU the same check is a device-side
        assertion that aborts the whole process instead of raising."""
        words = self.forward(z)
        batch_size, char_size = words.size(0), words.size(2)
        sample = torch.multinomial(words.detach().cpu().view(-1, char_size), 1)
        return sample.view(batch_size, max_length).to(words.device)


class _PoEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        M = mu.shape[0]
        n = mu[0].numel()
        mu, logvar = mu.contiguous().float(), logvar.contiguous().float()
        omu = torch.empty_like(mu[0]); olv = torch.empty_like(mu[0])
        call("mmvae_poe_fwd", ptr(mu), ptr(logvar), M, n, ptr(omu), ptr(olv), _stream())
        ctx.save_for_backward(mu, logvar)
        return omu, olv

    @staticmethod
    def backward(ctx, gmu, glv):
        mu, logvar = ctx.saved_tensors
        M, n = mu.shape[0], mu[0].numel()
        gmu = torch.zeros_like(mu[0]) if gmu is None else gmu.contiguous()
        glv = torch.zeros_like(mu[0]) if glv is None else glv.contiguous()
        dmu = torch.empty_like(mu); dlv = torch.empty_like(mu)
        call("mmvae_poe_bwd", ptr(mu), ptr(logvar), M, n, ptr(gmu), ptr(glv), ptr(dmu), ptr(dlv), _stream())
        return dmu, dlv


class ProductOfExperts(nn.Module):
    """multimnist/model.py:348-360 (variance-weighted mean, reproduced as written)."""
    def forward(self, mu, logvar, eps=1e-8):
        if eps != 1e-8:
            raise MMVAEError("ProductOfExperts: only eps=1e-8 (the reference default) is implemented")
        if mu.device.type != "cuda":
            raise MMVAEError("ProductOfExperts needs GPU tensors; there is no CPU fallback")
        _lib.init_device(mu.device.index if mu.device.index is not None else torch.cuda.current_device())
        return _PoEFn.apply(mu, logvar)


class _ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        z = torch.empty_like(mu)
        call("mmvae_reparam_fwd", ptr(mu), ptr(logvar), ptr(eps), mu.numel(), ptr(z), _stream())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = dz.contiguous()
        dmu = torch.empty_like(dz); dlv = torch.empty_like(dz)
        call("mmvae_reparam_bwd", ptr(logvar), ptr(eps), ptr(dz), dz.numel(), ptr(dmu), ptr(dlv), _stream())
        return dmu, dlv, None


class MultimodalVAE(nn.Module):
    """multimnist/model.py:21-93"""

    def __init__(self, n_latents=20, use_cuda=False):
        super().__init__()
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.text_encoder = TextEncoder(n_latents, n_characters, n_hiddens=100, bidirectional=True)
        self.text_decoder = TextDecoder(n_latents, n_characters, n_hiddens=100, use_cuda=use_cuda)
        self.experts = ProductOfExperts()
        self.n_latents = n_latents
        self._core = _Core(self, "", n_latents)
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
        mu = torch.zeros(size)
        logvar = torch.log(torch.ones(size))
        if use_cuda:
            mu, logvar = mu.cuda(), logvar.cuda()
        return mu, logvar

    def forward(self, image=None, text=None, eps=None, enc_masks=None, gru_keep=None, force_tokens=None):
        assert image is not None or text is not None
        if image is not None and text is not None:
            image_mu, image_logvar = self.image_encoder(image, enc_masks)
            text_mu, text_logvar = self.encode_text(text)
            mu = torch.stack((image_mu, text_mu), dim=0)
            logvar = torch.stack((image_logvar, text_logvar), dim=0)
        elif image is not None:
            mu, logvar = self.image_encoder(image, enc_masks)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        else:
            mu, logvar = self.encode_text(text)
            mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
        mu, logvar = self.experts(mu, logvar)
        z = self.reparametrize(mu, logvar, eps)
        image_recon = self.decode_image(z)
        text_recon = self.text_decoder(z, gru_keep, force_tokens)
        return image_recon, text_recon, mu, logvar


# ----------------------------------------------------------------------------------------------------------------
# loss_function (multimnist/train.py:69-87)
# ----------------------------------------------------------------------------------------------------------------
def _gscale(g: torch.Tensor) -> torch.Tensor:
    """The upstream gradient of a 0-d loss as a contiguous fp32 device scalar: the backward kernels read it themselves
    (a host ``.item()`` here would synchronise the stream three times per loss_function call)."""
    return g.detach().reshape(1).to(torch.float32).contiguous()


class _BCEMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t):
        p, t = p.contiguous().float(), t.contiguous().float()
        out = torch.zeros(1, dtype=torch.float32, device=p.device)
        call("mmvae_bce_fwd", ptr(p), ptr(t), p.numel(), ptr(out), _stream())
        ctx.save_for_backward(p, t)
        return (out / p.numel()).squeeze(0)

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        dp = torch.empty_like(p)
        call("mmvae_bce_bwd", ptr(p), ptr(t), p.numel(), 1.0 / p.numel(), ptr(_gscale(g)), ptr(dp), _stream())
        return dp, None


class _NLLMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target):
        logp, target = logp.contiguous().float(), target.contiguous().long()
        rows, classes = logp.shape
        out = torch.zeros(1, dtype=torch.float32, device=logp.device)
        call("mmvae_nll_fwd", ptr(logp), ptr(target), rows, classes, ptr(out), _stream())
        ctx.target, ctx.shape = target, (rows, classes)
        return (out / rows).squeeze(0)

    @staticmethod
    def backward(ctx, g):
        rows, classes = ctx.shape
        d = torch.empty(rows, classes, dtype=torch.float32, device=ctx.target.device)
        call("mmvae_nll_bwd", ptr(ctx.target), rows, classes, 1.0 / rows, ptr(_gscale(g)), ptr(d), _stream())
        return d, None


class _KLSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = mu.contiguous().float(), logvar.contiguous().float()
        out = torch.zeros(1, dtype=torch.float32, device=mu.device)
        call("mmvae_kl_fwd", ptr(mu), ptr(logvar), mu.numel(), ptr(out), _stream())
        ctx.save_for_backward(mu, logvar)
        return out.squeeze(0)

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        dmu = torch.empty_like(mu); dlv = torch.empty_like(mu)
        call("mmvae_kl_bwd", ptr(mu), ptr(logvar), mu.numel(), 1.0, ptr(_gscale(g)), ptr(dmu), ptr(dlv), _stream())
        return dmu, dlv


def loss_function(mu, logvar, recon_image=None, image=None, recon_text=None, text=None,
                  kl_lambda=1e-3, lambda_xy=1., lambda_yx=1.):
    """multimnist/train.py:69-87; returns a 0-d tensor supporting .backward() and .item()."""
    batch_size = mu.size(0)
    image_BCE, text_BCE = 0, 0
    if recon_image is not None and image is not None:
        image_BCE = lambda_xy * _BCEMeanFn.apply(recon_image.reshape(-1, 1 * 50 * 50), image.reshape(-1, 1 * 50 * 50))
    if recon_text is not None and text is not None:
        text_BCE = lambda_yx * _NLLMeanFn.apply(recon_text.reshape(-1, recon_text.size(2)), text.reshape(-1))
    KLD = _KLSumFn.apply(mu, logvar)
    KLD = KLD / batch_size * kl_lambda
    return image_BCE + text_BCE + KLD


elbo_loss = loss_function          # the name BASELINE.json uses


# ----------------------------------------------------------------------------------------------------------------
# fused trainer: the train() closure body of multimnist/train.py:146-173 as one enqueue
# ----------------------------------------------------------------------------------------------------------------
class FusedTrainer:
    """``FusedTrainer(vae, batch_size, lr)(image, text)`` == zero_grad + 3 passes + 3 losses + backward + Adam step,
    operating directly on ``vae``'s parameters (which stay ordinary nn.Parameters / state_dict entries)."""

    def __init__(self, vae: MultimodalVAE, batch_size: int, lr: float = 1e-3, kl_lambda: float = 1e-3, seed: int = 1234,
                 world_size: int = 1, all_reduce=None):
        dev = next(vae.parameters()).device
        self.vae = vae
        st = vae._core.sync(dev)
        self.engine = FusedELBOStep(st, batch_size, lr=lr, kl_lambda=kl_lambda, seed=seed, world_size=world_size,
                                    all_reduce=all_reduce)

    def __call__(self, image, text, **kw) -> StepOutputs:
        self.engine.kl_lambda = kw.pop("kl_lambda", self.engine.kl_lambda)
        self.engine.enc_dropout = self.vae.image_encoder.classifier[2].p > 0
        self.engine.gru_dropout = self.vae.text_decoder.gru.dropout > 0
        # the engine updates the flat parameter buffer in place and refreshes the packed bf16 copies itself
        return self.engine(image, text, **kw)

    def evaluate(self, image, text, **kw) -> StepOutputs:
        """The test() closure body (multimnist/train.py:197-209): eval-mode forward of the 3 passes, no backward."""
        return self.engine.forward_backward(image, text, training=False, backward=False, **kw)
