"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against
(1) the golden vectors captured from the reference and (2) the CPU oracle on the same seeded inputs.

Tolerances (stated, bf16 MFMA inputs / fp32 accumulation; fp32 everywhere else):
  ELBO losses rel 1e-3 (north-star bound) | mu/logvar abs 1e-2 | per-tensor grad rel-L2 3e-2 | total grad norm rel 1e-2.
The greedy text decoder feeds back argmax tokens (discontinuous): the parity runs force the reference's token path
(recorded in the fixture) and separately check that the free-running path agrees on >= 90 % of the decisions.
"""
import os

import numpy as np
import pytest
import torch

from oracle import mmvae_ref as R
from gradcheck import check_gradients

pytestmark = pytest.mark.gpu

D = 100


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _state_with_formula_params(dev):
    from multimodal_vae_amd.core import MultimnistState
    P = R.formula_params("multimnist", D, requires_grad=True)
    st = MultimnistState(D, dev)
    assert [t[0] for t in st.table] == [n for n, _ in R.param_table("multimnist", D)]
    for n, shape, off in st.table:
        assert tuple(P[n].shape) == tuple(shape)
        st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(dev)
    return st, P


NORM_TOL_B256 = 5e-3        # per-tensor gradient NORM of the free-running B=256 step, measured 1.6e-3 (direction: test_gpu_parity_r2.py)


def _unpack_masks(fx, B):
    m1 = np.stack([np.unpackbits(fx[f"mask_{i}_0"], axis=1)[:, :400] for i in range(2)])
    m2 = np.stack([np.unpackbits(fx[f"mask_{i}_1"], axis=1)[:, :200] for i in range(2)])
    return torch.from_numpy(m1.astype(np.uint8)), torch.from_numpy(m2.astype(np.uint8))


def _grad_checks(st, P, tot_tol=1e-3, tensor_tol=2.5e-2, label="multimnist"):
    g = st.grads.cpu()
    check_gradients(((n, g[off:off + P[n].numel()], P[n].grad) for n, shape, off in st.table), tensor_tol, tot_tol, label)


@pytest.mark.parametrize("fixture", ["multimnist_b8", "multimnist_b8_masks"])
def test_fused_step_matches_golden_and_oracle(fixture, golden_dir):
    from multimodal_vae_amd.core import FusedELBOStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, fixture + ".npz"))
    B = int(fx["B"])
    wm = bool(fx["with_masks"])
    st, P = _state_with_formula_params(dev)
    image, text = R.formula_inputs("multimnist", B)
    eps = [torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)]
    ft = torch.from_numpy(np.stack([fx[f"tokens_{k}"] for k in range(3)])).long()
    eng = FusedELBOStep(st, B)
    eng.enc_dropout, eng.gru_dropout = wm, False
    kw = {}
    em = None
    if wm:
        m1, m2 = _unpack_masks(fx, B)
        kw = dict(enc_mask1=m1.to(dev).contiguous(), enc_mask2=m2.to(dev).contiguous())
        em = ([m1[0].float(), m2[0].float()], [m1[1].float(), m2[1].float()], None)
    mu = torch.zeros(3, B, D, device=dev); lv = torch.zeros(3, B, D, device=dev)
    rt = torch.zeros(3, B, 4, 12, device=dev); ri = torch.zeros(3, B, 2500, device=dev)
    out = eng.forward_backward(image.to(dev), text.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                               force_tokens=ft.reshape(3 * B, 4).to(dev).contiguous(), mu=mu, logvar=lv, recon_text=rt,
                               recon_image=ri, **kw)
    losses = out.losses().cpu().numpy()
    np.testing.assert_allclose(losses, fx["loss"], rtol=1e-3)                      # golden: ELBO within 1e-3 relative
    for k in range(3):
        np.testing.assert_allclose(mu[k].cpu().numpy(), fx[f"mu_{k}"], atol=1e-2)
        np.testing.assert_allclose(lv[k].cpu().numpy(), fx[f"logvar_{k}"], atol=1e-2)
        np.testing.assert_allclose(rt[k].cpu().numpy(), fx[f"second_recon_{k}"], atol=2e-2)
        np.testing.assert_allclose(ri[k].double().sum().item(), fx[f"image_recon_stats_{k}"][0], rtol=2e-3)
    # oracle on the same inputs: every gradient tensor
    o_losses, _ = R.multimnist_step_losses(P, image, text, True, 1e-3, eps, em, None, [ft[0], ft[1], ft[2]],
                                           0.1 if wm else 0.0, 0.0)
    (o_losses[0] + o_losses[1] + o_losses[2]).backward()
    _grad_checks(st, P)
    np.testing.assert_allclose(st.grads.double().norm().item(), float(fx["total_grad_norm"]), rtol=1e-2)
    # BatchNorm running statistics after the 3 passes (fixture = reference buffers)
    for pre, c, off in st.bn_table:
        np.testing.assert_allclose(st.bn_stats[off:off + c].cpu().numpy(), fx["buf:" + pre + ".running_mean"], atol=2e-3)
        np.testing.assert_allclose(st.bn_stats[off + c:off + 2 * c].cpu().numpy(), fx["buf:" + pre + ".running_var"], rtol=2e-2, atol=1e-4)


def test_free_running_tokens_mostly_agree(golden_dir):
    from multimodal_vae_amd.core import FusedELBOStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "multimnist_b8.npz"))
    B = int(fx["B"])
    st, _ = _state_with_formula_params(dev)
    image, text = R.formula_inputs("multimnist", B)
    eps = torch.stack([torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)]).to(dev).contiguous()
    eng = FusedELBOStep(st, B)
    eng.enc_dropout = eng.gru_dropout = False
    tk = torch.zeros(3, B, 4, dtype=torch.int64, device=dev)
    out = eng.forward_backward(image.to(dev), text.to(dev), True, False, eps=eps, tokens=tk)
    ref = np.stack([fx[f"tokens_{k}"] for k in range(3)])
    agree = (tk.cpu().numpy() == ref).mean()
    assert agree >= 0.9, agree
    np.testing.assert_allclose(out.losses().cpu().numpy(), fx["loss"], rtol=5e-3)


def test_full_size_b256_scalars(golden_dir):
    """BASELINE.json's metric configuration (B=256): losses and per-tensor gradient norms of the reference."""
    from multimodal_vae_amd.core import FusedELBOStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "multimnist_b256_scalars.npz"))
    B = int(fx["B"])
    st, _ = _state_with_formula_params(dev)
    image, text = R.formula_inputs("multimnist", B)
    eps = []
    for k in range(3):
        torch.manual_seed(int(fx["seed0"]) + k)
        eps.append(torch.empty(B, D).normal_())
    eng = FusedELBOStep(st, B)
    eng.enc_dropout = eng.gru_dropout = False
    out = eng.forward_backward(image.to(dev), text.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous())
    # free-running greedy decode at full size (measured 2e-5; a near-tie flip of a fed-back token would show up here)
    if os.environ.get("MMVAE_TOL_REPORT"):
        print("TOL multimnist b256 free-running losses rel", np.abs(out.losses().cpu().numpy() / fx["loss"] - 1).max())
    np.testing.assert_allclose(out.losses().cpu().numpy(), fx["loss"], rtol=1e-3)
    g = st.grads.cpu()
    tot = float(fx["total_grad_norm"])
    np.testing.assert_allclose(g.double().norm().item(), tot, rtol=1e-3)           # measured 1.7e-4
    worst = 0.0
    for (n, shape, off), ref in zip(st.table, fx["grad_norms"]):
        numel = int(np.prod(shape))
        if ref < 1e-6 * tot:
            assert g[off:off + numel].double().norm().item() <= 1e-4 * tot, n
            continue
        rel = abs(g[off:off + numel].double().norm().item() - ref) / ref
        worst = max(worst, rel)
        assert rel <= NORM_TOL_B256, (n, rel)
    if os.environ.get("MMVAE_TOL_REPORT"):
        print("TOL multimnist b256 free-running: worst tensor-norm rel %.3e, total %.3e" % (worst, abs(g.double().norm().item() - tot) / tot))


def test_adam_matches_torch_semantics():
    from multimodal_vae_amd._lib import call, ptr
    from multimodal_vae_amd import _lib
    dev = _dev()
    _lib.init_device(0)
    n = 100003
    g0 = torch.Generator().manual_seed(3)
    p = torch.randn(n, generator=g0); g = torch.randn(n, generator=g0) * 0.1
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    s = __import__("ctypes").c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(3):
        gs = g * (step + 1)
        ref.grad = gs.clone()
        opt.step()
        call("mmvae_adam_step", ptr(pd), ptr(gs.to(dev)), ptr(m), ptr(v), n, ptr(state), 1e-3, 0.9, 0.999, 1e-8, 1.0, s)
    torch.cuda.synchronize()
    assert int(state[0].item()) == 3
    np.testing.assert_allclose(pd.cpu().numpy(), ref.detach().numpy(), atol=2e-6)


def test_packed_adam_equals_unpack_then_adam():
    """mmvae_adam_step_packed (gradient gathered from the packed weight-gradient buffers through mmvae_mm_grad_map) ==
    unpack kernel + mmvae_adam_step, bit for bit, on the same buffers; and the flat gradient is complete afterwards."""
    from multimodal_vae_amd._lib import call, ptr
    from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    B = 16
    st = MultimnistState(D, dev); default_init_(st, 5)
    eng = FusedELBOStep(st, B, seed=3)
    image, text = R.formula_inputs("multimnist", B)
    eng.forward_backward(image.to(dev).contiguous(), text.to(dev).contiguous(), True, True, _defer_unpack=True)
    gmap = st.grad_map()
    assert int((gmap >= 0).sum()) > 0.9 * st.nparams and int((gmap < -1).sum()) == 0 and int((gmap == -1).sum()) > 0
    used = gmap[gmap >= 0]
    assert used.unique().numel() == used.numel()                         # every packed element feeds exactly one parameter
    direct, gpk = st.grads.clone(), st.gpk.clone()
    p0 = st.params.clone()
    s = __import__("ctypes").c_void_p(torch.cuda.current_stream().cuda_stream)
    # reference order: unpack, then Adam
    full = direct.clone()
    full[gmap >= 0] += gpk[gmap[gmap >= 0].long()]
    pa, ma, va, sa = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), torch.zeros(2, dtype=torch.int64, device=dev)
    call("mmvae_adam_step", ptr(pa), ptr(full), ptr(ma), ptr(va), st.nparams, ptr(sa), 1e-3, 0.9, 0.999, 1e-8, 1.0, s)
    pb, gb, mb, vb, sb = p0.clone(), direct.clone(), torch.zeros_like(p0), torch.zeros_like(p0), torch.zeros(2, dtype=torch.int64, device=dev)
    call("mmvae_adam_step_packed", ptr(pb), ptr(gb), ptr(mb), ptr(vb), st.nparams, ptr(sb), 1e-3, 0.9, 0.999, 1e-8, 1.0,
         ptr(gmap), ptr(gpk), ptr(st.gpk_vec), s)
    torch.cuda.synchronize()
    assert torch.equal(gb, full) and torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    # and the step-level switch: fused __call__ leaves the same complete gradient as forward_backward's own unpack
    eng2 = FusedELBOStep(st, B, seed=3)
    eps = torch.stack([R.formula_eps(B, D, k) for k in range(3)]).to(dev).contiguous()
    eng2.enc_dropout = eng2.gru_dropout = False
    eng2.forward_backward(image.to(dev).contiguous(), text.to(dev).contiguous(), True, True, eps=eps)
    g_ref = st.grads.clone()
    eng2(image.to(dev).contiguous(), text.to(dev).contiguous(), eps=eps)
    assert ((st.grads - g_ref).norm() / g_ref.norm()).item() < 2e-3      # (two runs differ by the atomics' summation order)


def test_training_reduces_loss_and_graph_matches_eager():
    """Size-independent properties at the full configuration: loss goes down; a captured HIP graph replays the same step."""
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    B = 256
    rng = np.random.default_rng(0)
    img = torch.from_numpy((rng.random((B, 1, 50, 50), dtype=np.float32) * (rng.random((B, 1, 50, 50)) < 0.15)).astype(np.float32)).to(dev)
    txt = torch.from_numpy(rng.integers(0, 10, size=(B, 4)).astype(np.int64)).to(dev)
    st = MultimnistState(D, dev); default_init_(st, 7)
    eng = FusedELBOStep(st, B)
    first = eng(img, txt).losses().sum().item()
    for _ in range(30):
        last = eng(img, txt).losses().sum().item()
    assert np.isfinite(last) and last < 0.8 * first, (first, last)
    eng.capture(img, txt)
    a = eng.replay().losses().sum().item()
    b = eng.replay().losses().sum().item()
    assert np.isfinite(a) and np.isfinite(b) and b < first
    # the replayed graph holds EVERY kernel of the step (side streams included: a fork that is not a capture node leaves the weight
    # gradients and the text path out of the graph and the replay trains the image chain only): twin engines from one
    # initialisation, six steps eager against 2 eager (capture's warm-up) + 4 replays -- same device step counters, same draws
    sa = MultimnistState(D, dev); default_init_(sa, 11)
    sb = MultimnistState(D, dev); default_init_(sb, 11)
    p0 = sa.params.clone()
    ea, eb = FusedELBOStep(sa, B, seed=5), FusedELBOStep(sb, B, seed=5)
    for _ in range(6):
        la = ea(img, txt).losses().cpu().numpy()
    eb.capture(img, txt)
    for _ in range(4):
        lb = eb.replay().losses().cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(lb, la, rtol=2e-2)
    moved = (sa.params - p0).norm().item()
    assert (sa.params - sb.params).norm().item() < 0.1 * moved, ((sa.params - sb.params).norm().item(), moved)
    for n, shape, off in sa.table:                               # per tensor: nothing is left behind
        k = int(np.prod(shape))
        da, db = sa.params[off:off + k] - p0[off:off + k], sb.params[off:off + k] - p0[off:off + k]
        if da.norm().item() > 1e-3 * moved:
            assert (da - db).norm().item() < 0.35 * da.norm().item(), n


def test_dropin_modules_match_oracle(golden_dir):
    """Reference-style loop through the drop-in model.py surface + loss_function + torch.optim.Adam."""
    from multimodal_vae_amd import multimnist as M
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "multimnist_b8.npz"))
    B = int(fx["B"])
    P = R.formula_params("multimnist", D, requires_grad=True)
    vae = M.MultimodalVAE(D, use_cuda=True)
    vae.load_state_dict({k: v.detach().clone() for k, v in P.items()}, strict=True)
    vae.cuda().train()
    vae.image_encoder.classifier[2].p = 0.0; vae.image_encoder.classifier[5].p = 0.0; vae.text_decoder.gru.dropout = 0.0
    image, text = R.formula_inputs("multimnist", B)
    imd, txd = image.to(dev), text.to(dev)
    eps = [torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)]
    ft = [torch.from_numpy(fx[f"tokens_{k}"]).long() for k in range(3)]
    opt = torch.optim.Adam(vae.parameters(), lr=1e-3)
    opt.zero_grad()
    args = ((imd, txd), (imd, None), (None, txd))
    total = 0
    for k in range(3):
        ri, rt, mu, lv = vae(image=args[k][0], text=args[k][1], eps=eps[k].to(dev), force_tokens=ft[k].to(dev))
        assert ri.shape == (B, 1, 50, 50) and rt.shape == (B, 4, 12) and mu.shape == (B, D)
        lxy, lyx = R.MULTIMNIST_LAMBDAS[k]
        l = M.loss_function(mu, lv, recon_image=ri, image=imd, recon_text=rt, text=txd, kl_lambda=1e-3, lambda_xy=lxy, lambda_yx=lyx)
        assert l.dim() == 0
        np.testing.assert_allclose(l.item(), fx["loss"][k], rtol=1e-3)
        total = total + l
    total.backward()
    o_losses, _ = R.multimnist_step_losses(P, image, text, True, 1e-3, eps, None, None, ft, 0.0, 0.0)
    (o_losses[0] + o_losses[1] + o_losses[2]).backward()
    for n, p in vae.named_parameters():
        gr, gh = P[n].grad, p.grad.cpu()
        assert (gh - gr).norm().item() <= 3e-2 * gr.norm().item() + 1e-3, n
    opt.step()
    ri, rt, mu, lv = vae(image=imd, text=txd)                    # after the update: repacked weights, still finite
    assert torch.isfinite(ri).all() and torch.isfinite(rt).all()
    # eval mode: running statistics, no dropout, z = mu (multimnist/model.py:38-39)
    vae.eval()
    ri, rt, mu, lv = vae(image=imd, text=txd)
    Pe = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
    with torch.no_grad():
        o = R.multimnist_forward(Pe, image, text, False)
    np.testing.assert_allclose(ri.detach().cpu().numpy(), o[0].numpy(), atol=5e-3)
    np.testing.assert_allclose(mu.detach().cpu().numpy(), o[2].numpy(), atol=1e-2)
    with pytest.raises(AssertionError):
        vae()                                                    # model.py:64


def test_no_cpu_fallback():
    from multimodal_vae_amd import multimnist as M, MMVAEError
    vae = M.MultimodalVAE(D)
    with pytest.raises(MMVAEError):
        vae(image=torch.zeros(2, 1, 50, 50), text=torch.zeros(2, 4, dtype=torch.long))


@pytest.mark.parametrize("B", [12, 20])
def test_fused_step_at_batches_not_a_multiple_of_8(B):
    """Batch sizes that are multiples of 4 but not of 8 (round-3 ADVICE: the fused BatchNorm staging has kernels compiled for 8
    images per workgroup only and no fallback; such batches now run the unfused chain): losses and every gradient tensor against the
    oracle on the same inputs, forced to the oracle's greedy tokens."""
    from multimodal_vae_amd.core import FusedELBOStep
    dev = _dev()
    st, P = _state_with_formula_params(dev)
    image, text = R.formula_inputs("multimnist", B)
    g = torch.Generator().manual_seed(77 + B)
    eps = [torch.randn(B, D, generator=g) for _ in range(3)]
    o_losses, o_outs = R.multimnist_step_losses(P, image, text, True, 1e-3, eps, None, None, None, 0.0, 0.0)
    ft = torch.stack([o[1].detach().argmax(-1) for o in o_outs]).long()          # the oracle's greedy path (B,4) per pass
    (o_losses[0] + o_losses[1] + o_losses[2]).backward()
    eng = FusedELBOStep(st, B)
    eng.enc_dropout = eng.gru_dropout = False
    out = eng.forward_backward(image.to(dev), text.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                               force_tokens=ft.reshape(3 * B, 4).to(dev).contiguous())
    np.testing.assert_allclose(out.losses().cpu().numpy(), np.array([float(x.detach()) for x in o_losses]), rtol=1e-3)
    _grad_checks(st, P, tot_tol=2e-3, tensor_tol=4e-2, label="multimnist b%d" % B)


@pytest.mark.parametrize("staged", [1, 0])
def test_step_prologue_refreshes_every_weight_copy(staged):
    """The fused step refreshes the bf16 GEMM copies of the weights itself after an optimizer step -- in stages: what the image
    encoder's forward reads in the prologue on the main stream, the text copies and the decoders' / backward-only copies on the
    second-modality stream (csrc/multimnist.hip build_plan, mm_step_body).  A copy left out would be one optimizer step stale, silently:
    after a step the whole packed buffer must equal a full pack of the parameters the step started from, bit for bit."""
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    from multimodal_vae_amd.init import default_init_
    from multimodal_vae_amd._lib import call
    dev = _dev()
    B = 16
    rng = np.random.default_rng(3)
    img = torch.from_numpy((rng.random((B, 1, 50, 50)) < 0.2).astype(np.float32)).to(dev)
    txt = torch.from_numpy(rng.integers(0, 10, size=(B, 4)).astype(np.int64)).to(dev)
    st = MultimnistState(D, dev); default_init_(st, 21)
    eng = FusedELBOStep(st, B, lr=1e-2)           # a large step: every weight moves by more than a bf16 ulp
    call("mmvae_debug_set", b"mm_stage_begin", staged)
    try:
        eng(img, txt)
        for _ in range(2):
            assert st.pack_pending
            p_start = st.params.clone()
            eng(img, txt)                          # prologue: packs p_start; then forward, backward, Adam
            torch.cuda.synchronize()
            got, gotv = st.packed.clone(), st.packed_vec.clone()
            p_end = st.params.clone()
            assert not torch.equal(p_start, p_end)
            st.params.copy_(p_start); st.pack_weights(); torch.cuda.synchronize()
            ref, refv = st.packed.clone(), st.packed_vec.clone()
            st.params.copy_(p_end); st.pack_pending = True
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), int((got.view(torch.int16) != ref.view(torch.int16)).sum())
            assert torch.equal(gotv, refv)
            st.params.copy_(p_end); st.pack_weights(); torch.cuda.synchronize()      # and it is not the pack of the NEW parameters
            assert not torch.equal(st.packed.view(torch.int16), got.view(torch.int16))
            st.pack_pending = True
    finally:
        call("mmvae_debug_set", b"mm_stage_begin", 1)


def test_early_optimizer_part_equals_one_adam_launch():
    """mmvae_mm_step_io.early_adam: the Adam update of image_decoder.* / text_decoder.* issued inside the step (on the weight-gradient
    stream, beside the encoders' backward) + mmvae_adam_step_packed_ranges over the encoders behind the step is torch.optim.Adam
    (multimnist/train.py:129,173) on the step's own gradient: every range updated exactly once, one step counted."""
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    B = 16
    rng = np.random.default_rng(5)
    img = torch.from_numpy((rng.random((B, 1, 50, 50)) < 0.2).astype(np.float32)).to(dev)
    txt = torch.from_numpy(rng.integers(0, 10, size=(B, 4)).astype(np.int64)).to(dev)
    for early in (True, False):
        st = MultimnistState(D, dev); default_init_(st, 31)
        eng = FusedELBOStep(st, B, seed=9)
        eng.early_adam = early
        lr, (b1, b2), eps = eng.lr, eng.betas, eng.eps
        m = torch.zeros_like(st.params, dtype=torch.float64); v = torch.zeros_like(m)
        for t in range(1, 4):
            p_before = st.params.double().clone()
            eng(img, txt)
            torch.cuda.synchronize()
            assert (eng._ea_ran.value == 1) if early else (getattr(eng, "_ea_ran", None) is None or eng._ea_ran.value == 0)
            g = st.grads.double()                  # the completed flat gradient of THIS step (the optimizer kernel wrote it back)
            m = b1 * m + (1 - b1) * g; v = b2 * v + (1 - b2) * g * g
            want = p_before - (lr / (1 - b1 ** t)) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps)
            assert int(eng.adam_state[0].item()) == t
            # (fp32 arithmetic of the kernel against float64 here: 1 - beta2 alone is 1.7e-4 off in fp32)
            assert float((eng.exp_avg.double() - m).abs().max()) <= 1e-4 * float(m.abs().max())
            assert float((eng.exp_avg_sq.double() - v).abs().max()) <= 1e-3 * float(v.abs().max())
            moved = float((want - p_before).norm())
            assert float((st.params.double() - want).norm()) < 2e-3 * moved, (early, t, float((st.params.double() - want).norm()), moved)
            m, v = eng.exp_avg.double().clone(), eng.exp_avg_sq.double().clone()      # (follow the kernel's moments: no drift over the steps)
