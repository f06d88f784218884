"""CPU ORACLE (test infrastructure, never the product path).

A plain-PyTorch fp32 restatement of the reference's MMVAE ELBO hot path,
written functionally over a ``state_dict``-shaped parameter dict so that the
same tensors drive the oracle, the golden fixtures and the HIP engine.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product package
(``multimodal-vae_amd``) never does and raises if its HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(`/root/reference/<ds>/model.py`, `train.loss_function`) in the build
container and asserts this restatement agrees with it; the vectors it writes to
``tests/golden/`` pin it on the GPU box where the reference does not exist.

Every function cites the reference lines it restates (paths under
/root/reference).  Extra keyword hooks (``eps``, dropout masks, ``force_tokens``)
exist so stochastic pieces can be injected; defaults reproduce the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# multimnist/utils.py:14-19
MAX_LENGTH = 4
N_CHARACTERS = 12
SOS = 10
FILL = 11

BN_EPS = 1e-5        # torch.nn.BatchNorm default
BN_MOMENTUM = 0.1    # torch.nn.BatchNorm default
DROP_P = 0.1         # multimnist/model.py:175,178,230,262 ; celeba/model.py:116


# ----------------------------------------------------------------------------
# bf16 storage-contract emulation (test instrument, OFF by default = the reference's fp32 arithmetic)
# ----------------------------------------------------------------------------
# The HIP engine feeds its MFMA GEMMs bf16 operands (fp32 accumulate) and keeps layer activations / activation
# gradients in bf16.  With ``bf16_contract(True)`` the MLP models below apply the same roundings at the same places
# (GEMM operands, stored pre-BatchNorm tensors, stored activation gradients), everything else staying fp32, so a
# test can separate "rounding the engine is designed to do" from an implementation error.  ReLU makes the MNIST
# gradients discontinuous in those roundings, which is why this instrument exists.
_BF16_CONTRACT = False


class bf16_contract:
    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        global _BF16_CONTRACT
        self.prev, _BF16_CONTRACT = _BF16_CONTRACT, self.on

    def __exit__(self, *a):
        global _BF16_CONTRACT
        _BF16_CONTRACT = self.prev


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _q(x: Tensor) -> Tensor:
    """value stored / fed to a GEMM as bf16"""
    return _RoundFwd.apply(x) if _BF16_CONTRACT else x


def _qg(x: Tensor) -> Tensor:
    """gradient w.r.t. x stored as bf16"""
    return _RoundBwd.apply(x) if _BF16_CONTRACT else x


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """F.linear; under the bf16 contract both GEMM operands are rounded, the accumulation and bias stay fp32."""
    return F.linear(_q(x), _q(w), b)


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def swish(x: Tensor) -> Tensor:
    """multimnist/model.py:375-381"""
    return x * torch.sigmoid(x)


def batch_norm(x: Tensor, p: Params, prefix: str, training: bool) -> Tensor:
    """nn.BatchNorm1d/2d as used at multimnist/model.py:163,166,169,200,203,206.

    train: biased batch variance for normalisation, running stats updated with
    momentum 0.1 using the *unbiased* variance; eval: running stats.
    ``p`` buffers are updated in place (like the nn.Module)."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        n = x.numel() // x.shape[1]
        if n <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / (n - 1))
            if prefix + ".num_batches_tracked" in p:
                p[prefix + ".num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    xh = (_q(x) - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)   # _q: identity unless bf16_contract
    return xh * w.view(shape) + b.view(shape)


def dropout(x: Tensor, training: bool, mask: Optional[Tensor], p: float = DROP_P) -> Tensor:
    """nn.Dropout(p): keep-mask / (1-p) in train, identity in eval.
    ``mask`` (0/1, same shape) injects the keep-mask; None draws one."""
    if not training or p == 0.0:
        return x
    if mask is None:
        mask = (torch.rand_like(x) >= p).to(x.dtype)
    return x * mask / (1.0 - p)


def gru_cell(x: Tensor, h: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """One nn.GRU step (gate order r,z,n; h' = (1-z)*n + z*h)."""
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def product_of_experts(mu: Tensor, logvar: Tensor, eps: float = 1e-8) -> Tuple[Tensor, Tensor]:
    """multimnist/model.py:355-360 (identical in every dataset dir).
    NB variance-weighted mean -- reproduced as written."""
    var = torch.exp(logvar) + eps
    pd_mu = torch.sum(mu * var, dim=0) / torch.sum(var, dim=0)
    pd_var = 1 / torch.sum(1 / var, dim=0)
    return pd_mu, torch.log(pd_var)


def reparametrize(mu: Tensor, logvar: Tensor, training: bool, eps: Optional[Tensor]) -> Tensor:
    """multimnist/model.py:33-39"""
    if not training:
        return mu
    std = torch.exp(0.5 * logvar)
    if eps is None:
        eps = torch.empty_like(std).normal_()
    return eps * std + mu


def kl_sum(mu: Tensor, logvar: Tensor) -> Tensor:
    """multimnist/train.py:85"""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())


# ----------------------------------------------------------------------------
# MultiMNIST  (multimnist/model.py, multimnist/train.py)
# ----------------------------------------------------------------------------
def multimnist_image_encoder(p: Params, x: Tensor, training: bool,
                             masks: Optional[Sequence[Optional[Tensor]]] = None,
                             pre: str = "image_encoder.", drop_p: float = DROP_P) -> Tensor:
    """multimnist/model.py:150-188 -> (B, 2D) (mu | logvar)."""
    m = masks if masks is not None else (None, None)
    x = swish(F.conv2d(x, p[pre + "features.0.weight"], None, 2, 1))
    x = F.conv2d(x, p[pre + "features.2.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.3", training))
    x = F.conv2d(x, p[pre + "features.5.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.6", training))
    x = F.conv2d(x, p[pre + "features.8.weight"], None, 2, 0)
    x = swish(batch_norm(x, p, pre + "features.9", training))
    x = x.reshape(-1, 256 * 2 * 2)
    x = F.linear(x, p[pre + "classifier.0.weight"], p[pre + "classifier.0.bias"])
    x = dropout(swish(x), training, m[0], drop_p)
    x = F.linear(x, p[pre + "classifier.3.weight"], p[pre + "classifier.3.bias"])
    x = dropout(swish(x), training, m[1], drop_p)
    return F.linear(x, p[pre + "classifier.6.weight"], p[pre + "classifier.6.bias"])


def multimnist_image_decoder_logits(p: Params, z: Tensor, training: bool,
                                    pre: str = "image_decoder.") -> Tensor:
    """multimnist/model.py:191-215 without the final sigmoid."""
    x = swish(F.linear(z, p[pre + "upsample.0.weight"], p[pre + "upsample.0.bias"]))
    x = x.view(-1, 256, 2, 2)
    x = F.conv_transpose2d(x, p[pre + "hallucinate.0.weight"], None, 2, 0)
    x = swish(batch_norm(x, p, pre + "hallucinate.1", training))
    x = F.conv_transpose2d(x, p[pre + "hallucinate.3.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.4", training))
    x = F.conv_transpose2d(x, p[pre + "hallucinate.6.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.7", training))
    return F.conv_transpose2d(x, p[pre + "hallucinate.9.weight"], None, 2, 1)


def multimnist_image_decoder(p: Params, z: Tensor, training: bool) -> Tensor:
    """multimnist/model.py:211-216"""
    return torch.sigmoid(multimnist_image_decoder_logits(p, z, training))


def multimnist_text_encoder(p: Params, text: Tensor, pre: str = "text_encoder.") -> Tensor:
    """multimnist/model.py:237-247 -> (B, 2D).

    Bidirectional 1-layer GRU; ``x[-1]`` takes the LAST time step of both
    directions: the reverse direction has then seen only token T-1 from h=0.
    GRU dropout=0.1 is a no-op for a single layer."""
    H = 100
    emb = p[pre + "embed.weight"][text]            # (B, T, H)
    B, T, _ = emb.shape
    h = emb.new_zeros(B, H)
    for t in range(T):
        h = gru_cell(emb[:, t], h, p[pre + "gru.weight_ih_l0"], p[pre + "gru.weight_hh_l0"],
                     p[pre + "gru.bias_ih_l0"], p[pre + "gru.bias_hh_l0"])
    hb = gru_cell(emb[:, T - 1], emb.new_zeros(B, H),
                  p[pre + "gru.weight_ih_l0_reverse"], p[pre + "gru.weight_hh_l0_reverse"],
                  p[pre + "gru.bias_ih_l0_reverse"], p[pre + "gru.bias_hh_l0_reverse"])
    return F.linear(h + hb, p[pre + "h2p.weight"], p[pre + "h2p.bias"])


def multimnist_text_decoder(p: Params, z: Tensor, training: bool,
                            gru_masks: Optional[Sequence[Optional[Tensor]]] = None,
                            force_tokens: Optional[Tensor] = None,
                            pre: str = "text_decoder.", drop_p: float = DROP_P) -> Tuple[Tensor, Tensor]:
    """multimnist/model.py:268-307 -> (log-probs (B,4,12), greedy tokens (B,4)).

    2-layer GRU, inter-layer dropout 0.1 in train (``gru_masks[t]`` injects the
    keep-mask of step t; the reference cannot inject them, so fixtures use
    p=0 i.e. all-ones masks).  Greedy argmax feedback, no teacher forcing;
    ``force_tokens`` (B,4) overrides the fed-back tokens (test hook)."""
    B = z.shape[0]
    c_in = torch.full((B,), SOS, dtype=torch.long)
    h0 = F.linear(z, p[pre + "z2h.weight"], p[pre + "z2h.bias"])
    h = [h0, h0]
    words, toks = [], []
    for i in range(MAX_LENGTH):
        x = torch.cat((swish(p[pre + "embed.weight"][c_in]), z), dim=1)
        h[0] = gru_cell(x, h[0], p[pre + "gru.weight_ih_l0"], p[pre + "gru.weight_hh_l0"],
                        p[pre + "gru.bias_ih_l0"], p[pre + "gru.bias_hh_l0"])
        mid = h[0]
        if training:
            mid = dropout(mid, True, None if gru_masks is None else gru_masks[i], drop_p)
        h[1] = gru_cell(mid, h[1], p[pre + "gru.weight_ih_l1"], p[pre + "gru.weight_hh_l1"],
                        p[pre + "gru.bias_ih_l1"], p[pre + "gru.bias_hh_l1"])
        o = F.linear(torch.cat((h[1], z), dim=1), p[pre + "h2o.weight"], p[pre + "h2o.bias"])
        lp = F.log_softmax(o, dim=1)
        words.append(lp)
        tok = lp.argmax(dim=1)
        toks.append(tok)
        c_in = tok if force_tokens is None else force_tokens[:, i]
    return torch.stack(words, dim=1), torch.stack(toks, dim=1)


def multimnist_forward(p: Params, image: Optional[Tensor], text: Optional[Tensor], training: bool,
                       eps: Optional[Tensor] = None, enc_masks=None, gru_masks=None,
                       force_tokens: Optional[Tensor] = None,
                       enc_drop_p: float = DROP_P, gru_drop_p: float = DROP_P):
    """multimnist/model.py:62-93 -> (image_recon, text_recon, mu, logvar, tokens)."""
    assert image is not None or text is not None
    D = p["image_decoder.upsample.0.weight"].shape[1]
    mus, lvs = [], []
    if image is not None:
        o = multimnist_image_encoder(p, image, training, enc_masks, drop_p=enc_drop_p)
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    if text is not None:
        o = multimnist_text_encoder(p, text)
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    mu, logvar = product_of_experts(torch.stack(mus, 0), torch.stack(lvs, 0))
    z = reparametrize(mu, logvar, training, eps)
    image_recon = multimnist_image_decoder(p, z, training)
    text_recon, toks = multimnist_text_decoder(p, z, training, gru_masks, force_tokens, drop_p=gru_drop_p)
    return image_recon, text_recon, mu, logvar, toks


def multimnist_loss(mu, logvar, recon_image=None, image=None, recon_text=None, text=None,
                    kl_lambda=1e-3, lambda_xy=1., lambda_yx=1.):
    """multimnist/train.py:69-87"""
    B = mu.shape[0]
    image_bce, text_nll = 0, 0
    if recon_image is not None and image is not None:
        image_bce = lambda_xy * F.binary_cross_entropy(recon_image.reshape(-1, 2500), image.reshape(-1, 2500))
    if recon_text is not None and text is not None:
        text_nll = lambda_yx * F.nll_loss(recon_text.reshape(-1, recon_text.shape[2]), text.reshape(-1))
    return image_bce + text_nll + kl_sum(mu, logvar) / B * kl_lambda


MULTIMNIST_LAMBDAS = ((1.0, 1.0), (1.0, 0.5), (0.0, 1.0))   # multimnist/train.py:158-166


def multimnist_step_losses(p: Params, image: Tensor, text: Tensor, training: bool = True,
                           kl_lambda: float = 1e-3, eps: Optional[Sequence[Tensor]] = None,
                           enc_masks=None, gru_masks=None, force_tokens=None,
                           enc_drop_p: float = DROP_P, gru_drop_p: float = DROP_P):
    """Three passes of multimnist/train.py:154-166 -> ([loss_1, loss_2, loss_3], per-pass outputs)."""
    e = eps if eps is not None else (None, None, None)
    em = enc_masks if enc_masks is not None else (None, None, None)
    gm = gru_masks if gru_masks is not None else (None, None, None)
    ft = force_tokens if force_tokens is not None else (None, None, None)
    args = ((image, text), (image, None), (None, text))
    losses, outs = [], []
    for k in range(3):
        ri, rt, mu, lv, tk = multimnist_forward(p, args[k][0], args[k][1], training, e[k], em[k], gm[k], ft[k],
                                                enc_drop_p, gru_drop_p)
        lxy, lyx = MULTIMNIST_LAMBDAS[k]
        losses.append(multimnist_loss(mu, lv, ri, image, rt, text, kl_lambda, lxy, lyx))
        outs.append((ri, rt, mu, lv, tk))
    return losses, outs


# ----------------------------------------------------------------------------
# MNIST  (mnist/model.py:14-185, mnist/train.py:64-81)
# ----------------------------------------------------------------------------
def _mlp_bn_relu(p: Params, x: Tensor, pre: str, idx: Sequence[int], training: bool) -> Tensor:
    """Linear -> BatchNorm1d -> ReLU stacks of mnist/model.py:103-111,123-131."""
    for i in idx[:-1]:
        x = _qg(linear(x, p[f"{pre}net.{i}.weight"], p[f"{pre}net.{i}.bias"]))
        x = torch.relu(_qg(batch_norm(x, p, f"{pre}net.{i + 1}", training)))
    i = idx[-1]
    return _qg(linear(x, p[f"{pre}net.{i}.weight"], p[f"{pre}net.{i}.bias"]))


def mnist_forward(p: Params, image: Optional[Tensor], text: Optional[Tensor], training: bool,
                  eps: Optional[Tensor] = None):
    """mnist/model.py:53-84 -> (image_recon (B,784), text_recon (B,10) log-probs, mu, logvar)."""
    assert image is not None or text is not None
    D = p["image_decoder.net.0.weight"].shape[1]
    mus, lvs = [], []
    if image is not None:
        o = _mlp_bn_relu(p, image, "image_encoder.", (0, 3, 6), training)          # :99-118
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    if text is not None:
        x = _qg(_q(p["text_encoder.net.0.weight"][text]))                           # :136-153
        x = torch.relu(_qg(batch_norm(x, p, "text_encoder.net.1", training)))
        o = _qg(linear(x, p["text_encoder.net.3.weight"], p["text_encoder.net.3.bias"]))
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    mu, logvar = product_of_experts(torch.stack(mus, 0), torch.stack(lvs, 0))
    z = reparametrize(mu, logvar, training, eps)
    image_recon = torch.sigmoid(_mlp_bn_relu(p, z, "image_decoder.", (0, 3, 6), training))   # :121-133
    t = _mlp_bn_relu(p, z, "text_decoder.", (0, 3), training)                                 # :156-170
    return image_recon, F.log_softmax(t, dim=1), mu, logvar


def mnist_loss(mu, logvar, recon_image=None, image=None, recon_text=None, text=None,
               lambda_xy=1., lambda_yx=1.):
    """mnist/train.py:64-81 (KL divided by B*(784/3), true division)."""
    B = mu.shape[0]
    a, b = 0, 0
    if recon_image is not None and image is not None:
        a = lambda_xy * F.binary_cross_entropy(recon_image, image.reshape(-1, 784))
    if recon_text is not None and text is not None:
        b = lambda_yx * F.nll_loss(recon_text, text)
    return a + b + kl_sum(mu, logvar) / (B * (784 / 3))


def mnist_step_losses(p: Params, image: Tensor, text: Tensor, training: bool = True,
                      eps: Optional[Sequence[Tensor]] = None):
    """mnist/train.py:136-147 (all lambdas 1)."""
    e = eps if eps is not None else (None, None, None)
    args = ((image, text), (image, None), (None, text))
    losses, outs = [], []
    for k in range(3):
        ri, rt, mu, lv = mnist_forward(p, args[k][0], args[k][1], training, e[k])
        losses.append(mnist_loss(mu, lv, ri, image, rt, text))
        outs.append((ri, rt, mu, lv))
    return losses, outs


# ----------------------------------------------------------------------------
# CelebA  (celeba/model.py:14-57,91-196, celeba/train.py:60-81)
# ----------------------------------------------------------------------------
N_ATTRS = 18   # celeba/datasets.py:26-28


def celeba_image_encoder(p: Params, x: Tensor, training: bool, mask: Optional[Tensor] = None,
                         pre: str = "image_encoder.", drop_p: float = DROP_P) -> Tensor:
    """celeba/model.py:91-128"""
    x = swish(F.conv2d(x, p[pre + "features.0.weight"], None, 2, 1))
    x = F.conv2d(x, p[pre + "features.2.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.3", training))
    x = F.conv2d(x, p[pre + "features.5.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.6", training))
    x = F.conv2d(x, p[pre + "features.8.weight"], None, 1, 0)
    x = swish(batch_norm(x, p, pre + "features.9", training))
    x = x.reshape(-1, 256 * 5 * 5)
    x = F.linear(x, p[pre + "classifier.0.weight"], p[pre + "classifier.0.bias"])
    x = dropout(swish(x), training, mask, drop_p)
    return F.linear(x, p[pre + "classifier.3.weight"], p[pre + "classifier.3.bias"])


def celeba_image_decoder(p: Params, z: Tensor, training: bool, pre: str = "image_decoder.") -> Tensor:
    """celeba/model.py:131-161"""
    x = swish(F.linear(z, p[pre + "upsample.0.weight"], p[pre + "upsample.0.bias"]))
    x = x.view(-1, 256, 5, 5)
    x = F.conv_transpose2d(x, p[pre + "hallucinate.0.weight"], None, 1, 0)
    x = swish(batch_norm(x, p, pre + "hallucinate.1", training))
    x = F.conv_transpose2d(x, p[pre + "hallucinate.3.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.4", training))
    x = F.conv_transpose2d(x, p[pre + "hallucinate.6.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.7", training))
    return torch.sigmoid(F.conv_transpose2d(x, p[pre + "hallucinate.9.weight"], None, 2, 1))


def celeba_forward(p: Params, image: Optional[Tensor], attrs: Optional[Tensor], training: bool,
                   eps: Optional[Tensor] = None, enc_mask: Optional[Tensor] = None, enc_drop_p: float = DROP_P):
    """celeba/model.py:36-57 -> (image_recon, attrs_recon, mu, logvar)."""
    assert image is not None or attrs is not None
    D = p["image_decoder.upsample.0.weight"].shape[1]
    mus, lvs = [], []
    if image is not None:
        o = celeba_image_encoder(p, image, training, enc_mask, drop_p=enc_drop_p)
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    if attrs is not None:
        x = F.linear(attrs, p["attrs_encoder.net.0.weight"], p["attrs_encoder.net.0.bias"])   # :164-178
        x = swish(batch_norm(x, p, "attrs_encoder.net.1", training))
        o = F.linear(x, p["attrs_encoder.net.3.weight"], p["attrs_encoder.net.3.bias"])
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    mu, logvar = product_of_experts(torch.stack(mus, 0), torch.stack(lvs, 0))
    z = reparametrize(mu, logvar, training, eps)
    image_recon = celeba_image_decoder(p, z, training)
    x = F.linear(z, p["attrs_decoder.net.0.weight"], p["attrs_decoder.net.0.bias"])           # :181-196
    x = swish(batch_norm(x, p, "attrs_decoder.net.1", training))
    attrs_recon = torch.sigmoid(F.linear(x, p["attrs_decoder.net.3.weight"], p["attrs_decoder.net.3.bias"]))
    return image_recon, attrs_recon, mu, logvar


def celeba_loss(mu, logvar, recon_x=None, x=None, recon_y=None, y=None,
                kl_lambda=1e-3, lambda_x=1., lambda_y=1.):
    """celeba/train.py:60-81 (per-attribute BCE loop averaged over attributes)."""
    B = mu.shape[0]
    x_bce, y_bce = 0, 0
    if recon_x is not None and x is not None:
        x_bce = F.binary_cross_entropy(recon_x.reshape(-1, 3 * 64 * 64), x.reshape(-1, 3 * 64 * 64))
    if recon_y is not None and y is not None:
        y_bce = 0
        for i in range(y.shape[1]):
            y_bce = y_bce + F.binary_cross_entropy(recon_y[:, i], y[:, i])
        y_bce = y_bce / y.shape[1]
    return lambda_x * x_bce + lambda_y * y_bce + kl_sum(mu, logvar) / B * kl_lambda


def celeba_step_losses(p: Params, image: Tensor, attrs: Tensor, training: bool = True,
                       eps: Optional[Sequence[Tensor]] = None, enc_masks=None, enc_drop_p: float = DROP_P):
    """celeba/train.py:138-147 (all defaults: kl_lambda 1e-3, lambdas 1)."""
    e = eps if eps is not None else (None, None, None)
    em = enc_masks if enc_masks is not None else (None, None, None)
    args = ((image, attrs), (image, None), (None, attrs))
    losses, outs = [], []
    for k in range(3):
        ri, ra, mu, lv = celeba_forward(p, args[k][0], args[k][1], training, e[k], em[k], enc_drop_p)
        losses.append(celeba_loss(mu, lv, ri, image, ra, attrs))
        outs.append((ri, ra, mu, lv))
    return losses, outs


# ----------------------------------------------------------------------------
# COCO  (coco/model.py:22-90,147-312 ; coco/train.py:66-84,146-164)
# ----------------------------------------------------------------------------
COCO_MAX_WORDS = 102   # coco/utils.py:12-15 (100 words + SOS + EOS)
COCO_EMB = 300         # GloVe-840B vectors
COCO_HID = 200
COCO_LAMBDAS = ((1.0, 1.0), (1.0, 1.0), (0.0, 1.0))    # coco/train.py:152-164 (lambda_xy, lambda_yx)


def coco_image_encoder(p: Params, x: Tensor, training: bool, masks=None,
                       pre: str = "image_encoder.", drop_p: float = DROP_P) -> Tensor:
    """coco/model.py:155-187: 3x32x32, four k4 s2 p1 convolutions, classifier 2048-1024-256-2D with two Dropouts."""
    m1, m2 = masks if masks is not None else (None, None)
    x = swish(F.conv2d(x, p[pre + "features.0.weight"], None, 2, 1))
    x = F.conv2d(x, p[pre + "features.2.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.3", training))
    x = F.conv2d(x, p[pre + "features.5.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.6", training))
    x = F.conv2d(x, p[pre + "features.8.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "features.9", training))
    x = x.reshape(-1, 512 * 2 * 2)
    x = dropout(swish(F.linear(x, p[pre + "classifier.0.weight"], p[pre + "classifier.0.bias"])), training, m1, drop_p)
    x = dropout(swish(F.linear(x, p[pre + "classifier.3.weight"], p[pre + "classifier.3.bias"])), training, m2, drop_p)
    return F.linear(x, p[pre + "classifier.6.weight"], p[pre + "classifier.6.bias"])


def coco_image_decoder(p: Params, z: Tensor, training: bool, pre: str = "image_decoder.") -> Tensor:
    """coco/model.py:190-216"""
    x = swish(F.linear(z, p[pre + "upsample.0.weight"], p[pre + "upsample.0.bias"]))
    x = x.view(-1, 512, 2, 2)
    x = F.conv_transpose2d(x, p[pre + "hallucinate.0.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.1", training))
    x = F.conv_transpose2d(x, p[pre + "hallucinate.3.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.4", training))
    x = F.conv_transpose2d(x, p[pre + "hallucinate.6.weight"], None, 2, 1)
    x = swish(batch_norm(x, p, pre + "hallucinate.7", training))
    return torch.sigmoid(F.conv_transpose2d(x, p[pre + "hallucinate.9.weight"], None, 2, 1))


def coco_text_encoder(p: Params, text: Tensor, pre: str = "text_encoder.") -> Tensor:
    """coco/model.py:219-245: (B, T, 300) GloVe vectors -> (B, 2D).

    Bidirectional 1-layer GRU(300 -> 200); ``x[-1]`` is the LAST time step of both directions, so the reverse direction
    has seen only vector T-1 from h = 0 (same quirk as multimnist/model.py:242-245).  dropout=0.1 on a single-layer GRU
    is a no-op."""
    B, T, _ = text.shape
    h = text.new_zeros(B, COCO_HID)
    for t in range(T):
        h = gru_cell(text[:, t], h, p[pre + "gru.weight_ih_l0"], p[pre + "gru.weight_hh_l0"],
                     p[pre + "gru.bias_ih_l0"], p[pre + "gru.bias_hh_l0"])
    hb = gru_cell(text[:, T - 1], text.new_zeros(B, COCO_HID),
                  p[pre + "gru.weight_ih_l0_reverse"], p[pre + "gru.weight_hh_l0_reverse"],
                  p[pre + "gru.bias_ih_l0_reverse"], p[pre + "gru.bias_hh_l0_reverse"])
    return F.linear(h + hb, p[pre + "h2p.weight"], p[pre + "h2p.bias"])


def coco_text_decoder(p: Params, z: Tensor, training: bool, sos: Tensor, steps: int = COCO_MAX_WORDS,
                      gru_masks: Optional[Sequence[Optional[Tensor]]] = None,
                      pre: str = "text_decoder.", drop_p: float = DROP_P) -> Tensor:
    """coco/model.py:256-312 -> sentence (B, steps, 300).

    ``h = z2h(z)`` repeated for the 2 GRU layers; the first input is the GloVe vector of '<s>' (``sos``, a constant of
    the module, coco/model.py:271-272); every later input is the PREVIOUS OUTPUT VECTOR itself (:286) -- a differentiable
    feedback, unlike the argmax of the character models.  Each step: GRU(300+D -> 200, 2 layers, inter-layer dropout 0.1
    in train) on ``w_in || z``, then Linear(200+D -> 300) on ``h_top || z``.  The EOS pre-fill of ``sentence`` (:275-277)
    is overwritten at every position."""
    B = z.shape[0]
    w_in = sos.to(z.dtype).reshape(1, COCO_EMB).expand(B, COCO_EMB)
    h0 = F.linear(z, p[pre + "z2h.weight"], p[pre + "z2h.bias"])
    h = [h0, h0]
    out = []
    for i in range(steps):
        x = torch.cat((w_in, z), dim=1)
        h[0] = gru_cell(x, h[0], p[pre + "gru.weight_ih_l0"], p[pre + "gru.weight_hh_l0"],
                        p[pre + "gru.bias_ih_l0"], p[pre + "gru.bias_hh_l0"])
        mid = h[0]
        if training:
            mid = dropout(mid, True, None if gru_masks is None else gru_masks[i], drop_p)
        h[1] = gru_cell(mid, h[1], p[pre + "gru.weight_ih_l1"], p[pre + "gru.weight_hh_l1"],
                        p[pre + "gru.bias_ih_l1"], p[pre + "gru.bias_hh_l1"])
        w_out = F.linear(torch.cat((h[1], z), dim=1), p[pre + "h2o.weight"], p[pre + "h2o.bias"])
        out.append(w_out)
        w_in = w_out
    return torch.stack(out, dim=1)


def coco_forward(p: Params, image: Optional[Tensor], text: Optional[Tensor], training: bool, sos: Tensor,
                 eps: Optional[Tensor] = None, enc_masks=None, gru_masks=None,
                 enc_drop_p: float = DROP_P, gru_drop_p: float = DROP_P, steps: int = COCO_MAX_WORDS):
    """coco/model.py:60-90 -> (image_recon, text_recon, mu, logvar)."""
    assert image is not None or text is not None
    D = p["image_decoder.upsample.0.weight"].shape[1]
    mus, lvs = [], []
    if image is not None:
        o = coco_image_encoder(p, image, training, enc_masks, drop_p=enc_drop_p)
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    if text is not None:
        o = coco_text_encoder(p, text)
        mus.append(o[:, :D]); lvs.append(o[:, D:])
    mu, logvar = product_of_experts(torch.stack(mus, 0), torch.stack(lvs, 0))
    z = reparametrize(mu, logvar, training, eps)
    return (coco_image_decoder(p, z, training),
            coco_text_decoder(p, z, training, sos, steps, gru_masks, drop_p=gru_drop_p), mu, logvar)


def coco_loss(mu, logvar, recon_image=None, image=None, recon_text=None, text=None,
              kl_lambda=1e-3, lambda_xy=1., lambda_yx=1.):
    """coco/train.py:66-84: BCE mean over B*3*32*32, MSE mean over B*T*300, KL/B*kl_lambda."""
    B = mu.shape[0]
    image_bce, text_mse = 0, 0
    if recon_image is not None and image is not None:
        image_bce = lambda_xy * F.binary_cross_entropy(recon_image.reshape(-1, 3 * 32 * 32), image.reshape(-1, 3 * 32 * 32))
    if recon_text is not None and text is not None:
        text_mse = lambda_yx * F.mse_loss(recon_text, text)
    return image_bce + text_mse + kl_sum(mu, logvar) / B * kl_lambda


def coco_step_losses(p: Params, image: Tensor, text: Tensor, sos: Tensor, training: bool = True, kl_lambda: float = 1e-3,
                     eps: Optional[Sequence[Tensor]] = None, enc_masks=None, gru_masks=None,
                     enc_drop_p: float = DROP_P, gru_drop_p: float = DROP_P):
    """coco/train.py:146-165: passes (image,text), (image), (text) with COCO_LAMBDAS."""
    e = eps if eps is not None else (None, None, None)
    em = enc_masks if enc_masks is not None else (None, None, None)
    gm = gru_masks if gru_masks is not None else (None, None, None)
    args = ((image, text), (image, None), (None, text))
    T = text.shape[1]
    losses, outs = [], []
    for k in range(3):
        ri, rt, mu, lv = coco_forward(p, args[k][0], args[k][1], training, sos, e[k], em[k], gm[k],
                                      enc_drop_p, gru_drop_p, steps=T)
        losses.append(coco_loss(mu, lv, ri, image, rt, text, kl_lambda, COCO_LAMBDAS[k][0], COCO_LAMBDAS[k][1]))
        outs.append((ri, rt, mu, lv))
    return losses, outs


def formula_sos() -> Tensor:
    """Stand-in for GloVe('<s>') (the 2 GB GloVe file is not available anywhere in this build): a fixed 300-vector of
    GloVe-like scale.  The golden generator hands the same vector to the reference through its stubbed ``GloVe``."""
    return (0.4 * _hash_uniform(COCO_EMB, 999)).to(torch.float32)


# ----------------------------------------------------------------------------
# parameter tables (state_dict order and shapes, SURVEY 8 a14 [probed])
# ----------------------------------------------------------------------------
def _bn(name: str, c: int):
    return [(name + ".weight", (c,)), (name + ".bias", (c,))]


def _gru(name: str, inp: int, hid: int, suffixes: Sequence[str]):
    out = []
    for s in suffixes:
        i = inp if s in ("l0", "l0_reverse") else hid
        out += [(f"{name}.weight_ih_{s}", (3 * hid, i)), (f"{name}.weight_hh_{s}", (3 * hid, hid)),
                (f"{name}.bias_ih_{s}", (3 * hid,)), (f"{name}.bias_hh_{s}", (3 * hid,))]
    return out


def param_table(model: str, D: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every nn.Parameter in ``MultimodalVAE(D).state_dict()`` order."""
    if model == "multimnist":
        t = [("image_encoder.features.0.weight", (32, 1, 4, 4)),
             ("image_encoder.features.2.weight", (64, 32, 4, 4))] + _bn("image_encoder.features.3", 64)
        t += [("image_encoder.features.5.weight", (128, 64, 4, 4))] + _bn("image_encoder.features.6", 128)
        t += [("image_encoder.features.8.weight", (256, 128, 4, 4))] + _bn("image_encoder.features.9", 256)
        t += [("image_encoder.classifier.0.weight", (400, 1024)), ("image_encoder.classifier.0.bias", (400,)),
              ("image_encoder.classifier.3.weight", (200, 400)), ("image_encoder.classifier.3.bias", (200,)),
              ("image_encoder.classifier.6.weight", (2 * D, 200)), ("image_encoder.classifier.6.bias", (2 * D,))]
        t += [("image_decoder.upsample.0.weight", (1024, D)), ("image_decoder.upsample.0.bias", (1024,)),
              ("image_decoder.hallucinate.0.weight", (256, 128, 4, 4))] + _bn("image_decoder.hallucinate.1", 128)
        t += [("image_decoder.hallucinate.3.weight", (128, 64, 4, 4))] + _bn("image_decoder.hallucinate.4", 64)
        t += [("image_decoder.hallucinate.6.weight", (64, 32, 5, 5))] + _bn("image_decoder.hallucinate.7", 32)
        t += [("image_decoder.hallucinate.9.weight", (32, 1, 4, 4))]
        t += [("text_encoder.embed.weight", (12, 100))] + _gru("text_encoder.gru", 100, 100, ("l0", "l0_reverse"))
        t += [("text_encoder.h2p.weight", (2 * D, 100)), ("text_encoder.h2p.bias", (2 * D,))]
        t += [("text_decoder.embed.weight", (12, 100)),
              ("text_decoder.z2h.weight", (100, D)), ("text_decoder.z2h.bias", (100,))]
        t += _gru("text_decoder.gru", 100 + D, 100, ("l0", "l1"))
        t += [("text_decoder.h2o.weight", (12, 100 + D)), ("text_decoder.h2o.bias", (12,))]
        return t
    if model == "mnist":
        t = []
        for pre, dims in (("image_encoder", (784, 400, 200, 2 * D)), ("image_decoder", (D, 200, 400, 784))):
            t += [(f"{pre}.net.0.weight", (dims[1], dims[0])), (f"{pre}.net.0.bias", (dims[1],))] + _bn(f"{pre}.net.1", dims[1])
            t += [(f"{pre}.net.3.weight", (dims[2], dims[1])), (f"{pre}.net.3.bias", (dims[2],))] + _bn(f"{pre}.net.4", dims[2])
            t += [(f"{pre}.net.6.weight", (dims[3], dims[2])), (f"{pre}.net.6.bias", (dims[3],))]
        # state_dict order: image_encoder, image_decoder, text_encoder, text_decoder (mnist/model.py:17-20)
        t += [("text_encoder.net.0.weight", (10, 50))] + _bn("text_encoder.net.1", 50)
        t += [("text_encoder.net.3.weight", (2 * D, 50)), ("text_encoder.net.3.bias", (2 * D,))]
        t += [("text_decoder.net.0.weight", (10, D)), ("text_decoder.net.0.bias", (10,))] + _bn("text_decoder.net.1", 10)
        t += [("text_decoder.net.3.weight", (10, 10)), ("text_decoder.net.3.bias", (10,))]
        return t
    if model == "celeba":
        t = [("image_encoder.features.0.weight", (32, 3, 4, 4)),
             ("image_encoder.features.2.weight", (64, 32, 4, 4))] + _bn("image_encoder.features.3", 64)
        t += [("image_encoder.features.5.weight", (128, 64, 4, 4))] + _bn("image_encoder.features.6", 128)
        t += [("image_encoder.features.8.weight", (256, 128, 4, 4))] + _bn("image_encoder.features.9", 256)
        t += [("image_encoder.classifier.0.weight", (1024, 6400)), ("image_encoder.classifier.0.bias", (1024,)),
              ("image_encoder.classifier.3.weight", (2 * D, 1024)), ("image_encoder.classifier.3.bias", (2 * D,))]
        t += [("image_decoder.upsample.0.weight", (6400, D)), ("image_decoder.upsample.0.bias", (6400,)),
              ("image_decoder.hallucinate.0.weight", (256, 128, 4, 4))] + _bn("image_decoder.hallucinate.1", 128)
        t += [("image_decoder.hallucinate.3.weight", (128, 64, 4, 4))] + _bn("image_decoder.hallucinate.4", 64)
        t += [("image_decoder.hallucinate.6.weight", (64, 32, 4, 4))] + _bn("image_decoder.hallucinate.7", 32)
        t += [("image_decoder.hallucinate.9.weight", (32, 3, 4, 4))]
        t += [("attrs_encoder.net.0.weight", (64, N_ATTRS)), ("attrs_encoder.net.0.bias", (64,))] + _bn("attrs_encoder.net.1", 64)
        t += [("attrs_encoder.net.3.weight", (2 * D, 64)), ("attrs_encoder.net.3.bias", (2 * D,))]
        t += [("attrs_decoder.net.0.weight", (64, D)), ("attrs_decoder.net.0.bias", (64,))] + _bn("attrs_decoder.net.1", 64)
        t += [("attrs_decoder.net.3.weight", (N_ATTRS, 64)), ("attrs_decoder.net.3.bias", (N_ATTRS,))]
        return t
    if model == "coco":
        t = [("image_encoder.features.0.weight", (64, 3, 4, 4)),
             ("image_encoder.features.2.weight", (128, 64, 4, 4))] + _bn("image_encoder.features.3", 128)
        t += [("image_encoder.features.5.weight", (256, 128, 4, 4))] + _bn("image_encoder.features.6", 256)
        t += [("image_encoder.features.8.weight", (512, 256, 4, 4))] + _bn("image_encoder.features.9", 512)
        t += [("image_encoder.classifier.0.weight", (1024, 2048)), ("image_encoder.classifier.0.bias", (1024,)),
              ("image_encoder.classifier.3.weight", (256, 1024)), ("image_encoder.classifier.3.bias", (256,)),
              ("image_encoder.classifier.6.weight", (2 * D, 256)), ("image_encoder.classifier.6.bias", (2 * D,))]
        t += [("image_decoder.upsample.0.weight", (2048, D)), ("image_decoder.upsample.0.bias", (2048,)),
              ("image_decoder.hallucinate.0.weight", (512, 256, 4, 4))] + _bn("image_decoder.hallucinate.1", 256)
        t += [("image_decoder.hallucinate.3.weight", (256, 128, 4, 4))] + _bn("image_decoder.hallucinate.4", 128)
        t += [("image_decoder.hallucinate.6.weight", (128, 64, 4, 4))] + _bn("image_decoder.hallucinate.7", 64)
        t += [("image_decoder.hallucinate.9.weight", (64, 3, 4, 4))]
        t += _gru("text_encoder.gru", COCO_EMB, COCO_HID, ("l0", "l0_reverse"))
        t += [("text_encoder.h2p.weight", (2 * D, COCO_HID)), ("text_encoder.h2p.bias", (2 * D,))]
        t += [("text_decoder.z2h.weight", (COCO_HID, D)), ("text_decoder.z2h.bias", (COCO_HID,))]
        t += _gru("text_decoder.gru", COCO_EMB + D, COCO_HID, ("l0", "l1"))
        t += [("text_decoder.h2o.weight", (COCO_EMB, COCO_HID + D)), ("text_decoder.h2o.bias", (COCO_EMB,))]
        return t
    raise ValueError(model)


def bn_layers(model: str, D: int) -> List[Tuple[str, int]]:
    """(prefix, channels) of every BatchNorm, in state_dict order."""
    out = []
    tab = param_table(model, D)
    names = [n for n, _ in tab]
    for i, (n, s) in enumerate(tab):
        if n.endswith(".weight") and len(s) == 1 and n[:-7] + ".bias" in names:
            out.append((n[:-7], s[0]))
    return out


def _hash_uniform(n: int, salt: int) -> Tensor:
    """Exact-integer hash -> uniform(-1, 1) float64 (bitwise reproducible on every machine, no RNG state)."""
    x = (torch.arange(n, dtype=torch.int64) * 2654435761 + salt * 40503 + 12345) & 0xFFFFFFFF
    x = (x ^ (x >> 15)) * 2246822519 & 0xFFFFFFFF
    x = (x ^ (x >> 13)) * 3266489917 & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return (x.to(torch.float64) + 0.5) / 2147483648.0 - 1.0


def formula_params(model: str, D: int, requires_grad: bool = False) -> Params:
    """RNG-free initialisation shared by the golden generator, the oracle and the HIP tests.  It mimics the
    reference's default PyTorch init in distribution (Linear/Conv/GRU: U(+-1/sqrt(fan_in)), Embedding ~unit
    scale, BN weight near 1) but comes from an exact integer hash, so both sides regenerate identical tensors.
    Fresh BN buffers (mean 0, var 1, count 0)."""
    p: Params = {}
    bn = dict(bn_layers(model, D))
    for k, (name, shape) in enumerate(param_table(model, D)):
        n = 1
        for s_ in shape:
            n *= s_
        u = _hash_uniform(n, k + 1)
        prefix = name.rsplit(".", 1)[0]
        if prefix in bn and len(shape) == 1:
            v = 1.0 + 0.1 * u if name.endswith(".weight") else 0.1 * u
        elif "embed" in name or (model == "mnist" and name == "text_encoder.net.0.weight"):
            v = 1.7 * u                                          # ~unit variance like N(0,1)
        elif "gru" in name:
            v = u * (1.0 / math.sqrt(COCO_HID) if model == "coco" else 0.1)    # U(+-1/sqrt(hidden))
        elif len(shape) == 1:
            v = 0.05 * u
        else:
            if len(shape) == 4 and "hallucinate" in name:       # ConvTranspose (Cin, Cout, kh, kw): fan_in = Cout*kh*kw
                fan_in = shape[1] * shape[2] * shape[3]
            else:
                fan_in = n // shape[0]
            v = u / math.sqrt(fan_in)
        p[name] = v.to(torch.float32).reshape(shape).clone().requires_grad_(requires_grad)
    for prefix, c in bn.items():
        p[prefix + ".running_mean"] = torch.zeros(c)
        p[prefix + ".running_var"] = torch.ones(c)
        p[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return p


def formula_inputs(model: str, B: int, seed: int = 0):
    """RNG-free inputs: image in [0,1] (MultiMNIST-like sparsity), tokens/labels from an LCG."""
    def lcg(n, mod, s):
        out, x = [], (seed * 7919 + s) & 0x7FFFFFFF
        for _ in range(n):
            x = (1103515245 * x + 12345) & 0x7FFFFFFF
            out.append((x >> 16) % mod)
        return torch.tensor(out, dtype=torch.long)
    if model == "multimnist":
        i = torch.arange(B * 2500, dtype=torch.float64)
        img = (0.5 + 0.5 * torch.sin(0.011 * i * (1 + 0.001 * seed) + 0.3 * torch.sin(0.13 * i))).clamp(0, 1)
        img = torch.where(torch.sin(0.07 * i + seed) > 0.6, img, torch.zeros_like(img))
        text = lcg(B * 4, 10, 1).view(B, 4)
        lens = lcg(B, 5, 2)
        for b in range(B):
            text[b, int(lens[b]):] = FILL
        return img.to(torch.float32).view(B, 1, 50, 50), text
    if model == "mnist":
        i = torch.arange(B * 784, dtype=torch.float64)
        img = (0.5 + 0.5 * torch.sin(0.017 * i + seed)).clamp(0, 1)
        return img.to(torch.float32).view(B, 784), lcg(B, 10, 3)
    if model == "celeba":
        i = torch.arange(B * 3 * 64 * 64, dtype=torch.float64)
        img = (0.5 + 0.5 * torch.sin(0.0031 * i + seed)).clamp(0, 1)
        attrs = (lcg(B * N_ATTRS, 10, 4) < 3).to(torch.float32).view(B, N_ATTRS)
        return img.to(torch.float32).view(B, 3, 64, 64), attrs
    if model == "coco":
        i = torch.arange(B * 3 * 32 * 32, dtype=torch.float64)
        img = (0.5 + 0.5 * torch.sin(0.0071 * i + seed)).clamp(0, 1)
        # caption tensors (coco/utils.py:36-47): GloVe-like vectors for the first `len` positions, zeros after
        j = torch.arange(B * COCO_MAX_WORDS * COCO_EMB, dtype=torch.float64)
        txt = (0.4 * (torch.sin(0.37 * j + 0.11 * seed) + 0.5 * torch.cos(0.0131 * j))).view(B, COCO_MAX_WORDS, COCO_EMB)
        lens = 8 + lcg(B, 40, 5)
        for b in range(B):
            txt[b, int(lens[b]):] = 0.0
        return img.to(torch.float32).view(B, 3, 32, 32), txt.to(torch.float32)
    raise ValueError(model)


def sample_idx(n: int, k: int = 32):
    """Evenly spread sample positions of a flat tensor (fixtures store gradient samples at these positions)."""
    import numpy as np
    return np.unique(np.linspace(0, n - 1, num=min(k, n)).astype(np.int64))


def formula_eps(B: int, D: int, k: int) -> Tensor:
    """Deterministic stand-in for N(0,1) draws: a smooth, roughly unit-variance field."""
    i = torch.arange(B * D, dtype=torch.float64)
    return (1.4 * torch.sin(0.91 * i + 1.7 * k) + 0.3 * torch.cos(0.113 * i * (k + 1))).to(torch.float32).view(B, D)


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults; multimnist/train.py:129,173)
# ----------------------------------------------------------------------------
def adam_step(params: Sequence[Tensor], grads: Sequence[Tensor], m: Sequence[Tensor], v: Sequence[Tensor],
              step: int, lr: float = 1e-3, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place Adam update, ``step`` is 1-based."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    with torch.no_grad():
        for p_, g, m_, v_ in zip(params, grads, m, v):
            m_.mul_(b1).add_(g, alpha=1 - b1)
            v_.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
            p_.addcdiv_(m_, denom, value=-lr / bc1)
