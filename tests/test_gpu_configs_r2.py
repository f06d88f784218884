"""BASELINE.json configurations 3 and 5 at their full per-GPU sizes, against the reference's own numbers
(fixtures ``celeba_b512_scalars`` / ``coco_b128_scalars`` written by oracle/make_golden.py from the imported reference:
celeba/train.py:131-147, coco/train.py:138-173), and the RCCL face of the data-parallel step on one rank.

Gates: the three ELBO losses rel 1e-3 (north_star) | total gradient norm rel 1e-2 | every gradient tensor by direction on
64 recorded samples and by norm (5e-2; tensors below 1e-4 of the total norm carry rounding noise only) | BatchNorm
running statistics.
"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import mmvae_ref as R

pytestmark = pytest.mark.gpu

D = 100


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _load(st, P):
    for n, shape, off in st.table:
        st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(st.device)


def _eps(fx, B):
    out = []
    for k in range(3):
        torch.manual_seed(int(fx["seed0"]) + k)
        out.append(torch.empty(B, D).normal_())
    return torch.stack(out)


def _check(st, fx, out, tol=5e-2, tot_tol=1e-2, buffers=True):
    losses = out.losses().cpu().numpy()
    np.testing.assert_allclose(losses, fx["loss"], rtol=1e-3)
    tot = float(fx["total_grad_norm"])
    g = st.grads.cpu().double()
    np.testing.assert_allclose(g.norm().item(), tot, rtol=tot_tol)
    worst = 0.0
    for i, (n, shape, off) in enumerate(st.table):
        numel = int(np.prod(shape))
        ref_norm = float(fx["grad_norms"][i])
        if ref_norm <= 1e-4 * tot:
            continue
        gt = g[off:off + numel]
        idx = R.sample_idx(numel, 64)
        ref = torch.from_numpy(fx["grad_samples"][i, :len(idx)])
        err = (gt[idx] - ref).norm().item() / max(ref.norm().item(), 1e-30)
        nerr = abs(gt.norm().item() - ref_norm) / ref_norm
        worst = max(worst, err, nerr)
        assert err <= tol and nerr <= tol, (n, err, nerr)
    for pre, c, off in st.bn_table:
        np.testing.assert_allclose(st.bn_stats[off:off + c].cpu().numpy(), fx["buf:" + pre + ".running_mean"], atol=3e-3)
        if buffers:
            np.testing.assert_allclose(st.bn_stats[off + c:off + 2 * c].cpu().numpy(), fx["buf:" + pre + ".running_var"], rtol=2e-2, atol=1e-4)
    return float(np.abs(losses / fx["loss"] - 1).max()), worst


def test_celeba_b512_matches_reference_numbers(golden_dir):
    """config 3: CelebA 64x64, batch 512"""
    from multimodal_vae_amd.core import CelebaState, FusedCelebaStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "celeba_b512_scalars.npz"))
    B = int(fx["B"])
    assert B == 512
    st = CelebaState(D, dev); _load(st, R.formula_params("celeba", D))
    image, attrs = R.formula_inputs("celeba", B)
    eng = FusedCelebaStep(st, B)
    eng.enc_dropout = False
    out = eng.forward_backward(image.to(dev).contiguous(), attrs.to(dev).contiguous(), True, True, eps=_eps(fx, B).to(dev).contiguous())
    print("celeba b512: loss rel / worst tensor", _check(st, fx, out))


def test_coco_b128_t102_matches_reference_numbers(golden_dir):
    """config 5's per-GPU share (1024 over 8 GPUs): COCO 32x32 + 102-step GloVe captions, batch 128"""
    from multimodal_vae_amd.core import CocoState, FusedCocoStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "coco_b128_scalars.npz"))
    B = int(fx["B"])
    assert B == 128
    st = CocoState(D, dev, 102); _load(st, R.formula_params("coco", D))
    image, text = R.formula_inputs("coco", B)
    assert text.shape == (B, 102, 300)
    eng = FusedCocoStep(st, B, R.formula_sos())
    eng.enc_dropout = eng.gru_dropout = False
    out = eng.forward_backward(image.to(dev).contiguous(), text.to(dev).contiguous(), True, True, eps=_eps(fx, B).to(dev).contiguous())
    print("coco b128: loss rel / worst tensor", _check(st, fx, out))


def _coco_replay(fx, B, T, dev):
    """eps / classifier keep masks / GRU inter-layer keep masks of a *_dropout fixture in the step's layouts."""
    eps = torch.stack([torch.from_numpy(fx[f"eps_{k}"]) for k in range(3)])
    m1 = np.stack([np.unpackbits(fx[f"encmask_{k}_0"], axis=1)[:, :1024] for k in range(2)]).astype(np.uint8)
    m2 = np.stack([np.unpackbits(fx[f"encmask_{k}_1"], axis=1)[:, :256] for k in range(2)]).astype(np.uint8)
    gk = np.concatenate([np.unpackbits(fx[f"grukeep_{k}"], axis=2)[:, :, :200] for k in range(3)], axis=1).astype(np.uint8)   # (T, 3B, 200)
    assert gk.shape == (T, 3 * B, 200)
    return dict(eps=eps.to(dev).contiguous(), enc_mask1=torch.from_numpy(m1).to(dev).contiguous(),
                enc_mask2=torch.from_numpy(m2).to(dev).contiguous(), gru_keep=torch.from_numpy(gk).to(dev).contiguous())


def test_coco_default_train_mode_with_the_references_own_dropout_draws(golden_dir):
    """coco/train.py's default mode -- classifier Dropout(0.1) twice and the caption GRU's inter-layer dropout ON -- with the
    reference's own draws (fixture coco_b4_dropout: replayed by seed from the imported reference, oracle/make_golden.py) fed to
    the fused step: the reference's losses, gradient norms / samples of every tensor and BatchNorm buffers."""
    from multimodal_vae_amd.core import CocoState, FusedCocoStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "coco_b4_dropout.npz"))
    B, T = int(fx["B"]), 102
    st = CocoState(D, dev, T); _load(st, R.formula_params("coco", D))
    image, text = R.formula_inputs("coco", B)
    eng = FusedCocoStep(st, B, R.formula_sos())
    assert eng.enc_dropout and eng.gru_dropout                                   # the defaults
    out = eng.forward_backward(image.to(dev).contiguous(), text.to(dev).contiguous(), True, True, **_coco_replay(fx, B, T, dev))
    print("coco b4 dropout: loss rel / worst tensor", _check(st, fx, out, tol=6e-2, buffers=False))


def test_coco_b128_with_dropout_on_matches_the_oracle():
    """config 5's per-GPU share once more with every dropout ON: seeded keep masks fed to both the fused step and the oracle
    (the reference has no fixture at this size with dropout; the oracle is pinned to it at B=4 with dropout and at B=128 without)."""
    from multimodal_vae_amd.core import CocoState, FusedCocoStep
    dev = _dev()
    B, T = 128, 102
    P = R.formula_params("coco", D, requires_grad=True)
    st = CocoState(D, dev, T); _load(st, P)
    image, text = R.formula_inputs("coco", B)
    g = torch.Generator().manual_seed(77)
    eps = [torch.randn(B, D, generator=g) for _ in range(3)]
    m1 = (torch.rand(2, B, 1024, generator=g) >= 0.1)
    m2 = (torch.rand(2, B, 256, generator=g) >= 0.1)
    gk = (torch.rand(T, 3 * B, 200, generator=g) >= 0.1)
    eng = FusedCocoStep(st, B, R.formula_sos())
    out = eng.forward_backward(image.to(dev).contiguous(), text.to(dev).contiguous(), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                               enc_mask1=m1.to(torch.uint8).to(dev).contiguous(), enc_mask2=m2.to(torch.uint8).to(dev).contiguous(),
                               gru_keep=gk.to(torch.uint8).to(dev).contiguous())
    em = ([m1[0].float(), m2[0].float()], [m1[1].float(), m2[1].float()], None)
    gm = tuple([gk[t, k * B:(k + 1) * B].float() for t in range(T)] for k in range(3))
    o_losses, _ = R.coco_step_losses(P, image, text, R.formula_sos(), True, 1e-3, eps, em, gm, 0.1, 0.1)
    (o_losses[0] + o_losses[1] + o_losses[2]).backward()
    np.testing.assert_allclose(out.losses().cpu().numpy(), [l.item() for l in o_losses], rtol=1e-3)
    gh = st.grads.cpu()
    from gradcheck import check_gradients
    check_gradients(((n, gh[off:off + P[n].numel()], P[n].grad) for n, shape, off in st.table), 5e-2, 1e-2, "coco b128 dropout on")


def test_coco_b1024_one_workgroup_per_block_decoder(golden_dir):
    """config 5 on ONE GPU (B = 1024, T = 102): the caption decoder runs in its one-workgroup-per-row-block form there (192 row
    blocks do not fit a cluster launch).  The oracle is too slow at this size, so the batch is the B = 128 formula batch (and its
    eps) tiled 8 times with dropout off: batch statistics, the three losses and every gradient tensor equal the B = 128
    numbers of the reference (means over identical copies; only the unbiased running_var factor differs)."""
    from multimodal_vae_amd.core import CocoState, FusedCocoStep
    dev = _dev()
    fx = np.load(os.path.join(golden_dir, "coco_b128_scalars.npz"))
    B0, B = int(fx["B"]), 1024
    if torch.cuda.get_device_properties(dev).total_memory < 24 << 30:
        pytest.skip("needs ~10 GB of workspace")
    st = CocoState(D, dev, 102); _load(st, R.formula_params("coco", D))
    image, text = R.formula_inputs("coco", B0)
    rep = B // B0
    eng = FusedCocoStep(st, B, R.formula_sos())
    eng.enc_dropout = eng.gru_dropout = False
    out = eng.forward_backward(image.repeat(rep, 1, 1, 1).to(dev).contiguous(), text.repeat(rep, 1, 1).to(dev).contiguous(), True, True,
                               eps=_eps(fx, B0).repeat(1, rep, 1).to(dev).contiguous())
    print("coco b1024 (8 x b128): loss rel / worst tensor", _check(st, fx, out, buffers=False))


@pytest.mark.parametrize("family", ["celeba", "coco"])
def test_packed_adam_equals_unpack_then_adam_other_families(family):
    """The optimizer path the CelebA / COCO steps use by default (mmvae_adam_step_packed: gradients gathered from the packed
    weight-gradient buffers through mmvae_<family>_grad_map) == unpack + mmvae_adam_step, bit for bit, on the same buffers."""
    import ctypes
    from multimodal_vae_amd._lib import call, ptr
    from multimodal_vae_amd import core
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    B = 8
    if family == "celeba":
        st = core.CelebaState(D, dev); default_init_(st, 5)
        eng = core.FusedCelebaStep(st, B, seed=3)
    else:
        st = core.CocoState(D, dev, 6); default_init_(st, 5)
        eng = core.FusedCocoStep(st, B, R.formula_sos(), seed=3)
    a, b = R.formula_inputs(family, B)
    if family == "coco":
        b = b[:, :6].contiguous()
    eng.forward_backward(a.to(dev).contiguous(), b.to(dev).contiguous(), True, True, _defer_unpack=True)
    gmap = st.grad_map()
    used = gmap[gmap >= 0]
    assert used.numel() > 0.8 * st.nparams and used.unique().numel() == used.numel()
    direct, gpk, gvec = st.grads.clone(), st.gpk.clone(), st.gpk_vec.clone()
    full = direct.clone()
    full[gmap >= 0] += gpk[gmap[gmap >= 0].long()]
    vec = gmap < -1
    full[vec] += gvec[(-gmap[vec] - 2).long()]
    p0 = st.params.clone()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pa, ma, va, sa = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), torch.zeros(2, dtype=torch.int64, device=dev)
    call("mmvae_adam_step", ptr(pa), ptr(full), ptr(ma), ptr(va), st.nparams, ptr(sa), 1e-4, 0.9, 0.999, 1e-8, 1.0, s)
    pb, gb, mb, vb, sb = p0.clone(), direct.clone(), torch.zeros_like(p0), torch.zeros_like(p0), torch.zeros(2, dtype=torch.int64, device=dev)
    call("mmvae_adam_step_packed", ptr(pb), ptr(gb), ptr(mb), ptr(vb), st.nparams, ptr(sb), 1e-4, 0.9, 0.999, 1e-8, 1.0,
         ptr(gmap), ptr(gpk), ptr(gvec), s)
    torch.cuda.synchronize()
    assert torch.equal(gb, full) and torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert int(sa[0]) == 1 and int(sb[0]) == 1
    # a void step (skip word set by a timed-out exchange) leaves everything alone and keeps the step count
    sb[1] = 1 << 32                                            # the word behind the block ticket
    pc = pb.clone()
    call("mmvae_adam_step", ptr(pb), ptr(full), ptr(mb), ptr(vb), st.nparams, ptr(sb), 1e-4, 0.9, 0.999, 1e-8, 1.0, s)
    torch.cuda.synchronize()
    assert torch.equal(pb, pc) and int(sb[0]) == 1 and int(sb[1]) == 0
    # ... and so does the void mark in element 0 (a NaN with the payload 0x7fc0dead), also after it went through the SUM of an
    # all-reduce with another rank's finite value
    mark = torch.tensor([0x7FC0DEAD], dtype=torch.int32, device=dev).view(torch.float32)
    g0 = float(full[0])
    full[0:1] = mark + torch.tensor([0.25], device=dev)
    assert (int(full[0:1].view(torch.int32)) & 0x7FFFFFFF) == 0x7FC0DEAD     # NaN + x keeps the payload
    call("mmvae_adam_step", ptr(pb), ptr(full), ptr(mb), ptr(vb), st.nparams, ptr(sb), 1e-4, 0.9, 0.999, 1e-8, 1.0, s)
    torch.cuda.synchronize()
    assert torch.equal(pb, pc) and int(sb[0]) == 1
    # an ordinary NaN is a gradient value, not a mark: the update runs and the NaN shows in the parameter, as with torch.optim.Adam
    # (multimnist/train.py:173)
    full[0] = float("nan")
    call("mmvae_adam_step", ptr(pb), ptr(full), ptr(mb), ptr(vb), st.nparams, ptr(sb), 1e-4, 0.9, 0.999, 1e-8, 1.0, s)
    torch.cuda.synchronize()
    assert int(sb[0]) == 2 and bool(torch.isnan(pb[0])) and not bool(torch.isnan(pb[1:]).any())
    full[0] = g0


def test_single_rank_rccl_group_runs_the_data_parallel_step():
    """The data-parallel branch of the engine on real hardware once: RCCL (backend "nccl") process group of world size 1,
    GradAllReduce on the flat gradient, separate-unpack path, Adam with grad_scale = 1/world, bucketed variant included,
    and the overlapped variant (decoder gradient ranges all-reduced on a communication stream behind the step's
    early-gradient event while the encoders' backward runs, mmvae_mm_step_io.dp_split).
    The result must equal the single-GPU packed path bit for bit in the gradients and closely in the parameters."""
    import torch.distributed as dist
    from multimodal_vae_amd import dp
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":             # exported by the image / gpurun; dp refuses without it
        pytest.skip("HSA_ENABLE_IPC_MODE_LEGACY=0 is not exported: RCCL cannot share device memory across processes here")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        B = 64
        image, text = R.formula_inputs("multimnist", B)
        imd, txd = image.to(dev).contiguous(), text.to(dev).contiguous()
        eps = torch.stack([R.formula_eps(B, D, k) for k in range(3)]).to(dev).contiguous()
        results = []
        for mode in ("packed", "dp", "dp_bucketed", "dp_overlap"):
            st = MultimnistState(D, dev); default_init_(st, 11)
            p_init = st.params.clone()
            if mode == "packed":
                eng = FusedELBOStep(st, B, seed=5)
            else:
                ar = dp.GradAllReduce(bucket_bytes=(1 << 20) if mode == "dp_bucketed" else 0, force=True,    # world 1: still
                                      overlap=mode == "dp_overlap")                                          # issue the collective
                eng = FusedELBOStep(st, B, seed=5, world_size=1, all_reduce=ar)
                assert eng._dp_active()
                if mode == "dp_overlap":
                    # the decoders' parameter ranges go out early (two runs of the flat buffer), the encoders' late
                    early, late = eng.grad_ranges()
                    assert sum(n for _, n in early) + sum(n for _, n in late) == st.nparams
                    names = [n for n, _, _ in st.table]
                    n_dec = sum(int(np.prod(s)) for n, s, _ in st.table if n.startswith(("image_decoder.", "text_decoder.")))
                    assert sum(n for _, n in early) == n_dec and len(early) == 2 and len(late) == 2, (early, late, names[:3])
            eng.enc_dropout = eng.gru_dropout = False
            out = eng(imd, txd, eps=eps)
            torch.cuda.synchronize()
            g_first = st.grads.clone()       # same parameters in every mode: only the fp32 summation order may differ
            out = eng(imd, txd, eps=eps)
            torch.cuda.synchronize()
            results.append((st.params - p_init, st.grads.clone(), out.losses().cpu().numpy(), g_first))
        p0, g0, l0, f0 = results[0]
        for p, g, l, f in results[1:]:
            # first step, same parameters: run-to-run noise only (fp32 atomics order feeding bf16 roundings: measured 3.5e-4
            # of the norm) -- and every parameter tensor's gradient has arrived (a range left out would read 1.0)
            assert ((f - f0).norm() / f0.norm()).item() < 2e-3
            for name, shape, off in st.table:
                n = int(np.prod(shape))
                a, b = f[off:off + n], f0[off:off + n]
                assert ((a - b).norm() / b.norm().clamp_min(1e-12)).item() < 0.1, name
            np.testing.assert_allclose(l, l0, rtol=1e-3)     # second step: the first update differs by the atomics order
            assert ((g - g0).norm() / g0.norm()).item() < 5e-3           # (measured 2.1e-3 between two runs)
            # p = the two Adam updates: Adam's first steps are sign-like (lr * g / |g|), so the atomics-order noise of the
            # gradients flips the few elements whose gradient is ~0: measured 2.6e-2 of the update norm
            assert ((p - p0).norm() / p0.norm()).item() < 8e-2
        t = torch.ones(1 << 20, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert float(t.sum().item()) == float(1 << 20)
    finally:
        if created:
            dist.destroy_process_group()


def test_c_abi_collective_face_single_rank():
    """mmvae_comm_* (SURVEY 8b collective face): RCCL bound with dlopen, a communicator of one rank, the in-place SUM
    all-reduce of a flat fp32 buffer on the caller's stream, then the data-parallel update through mmvae_adam_step with
    grad_scale = 1/world -- the three calls a C++ host makes per step."""
    import ctypes as C
    from multimodal_vae_amd._lib import call, ptr
    dev = _dev()
    torch.cuda.set_device(dev)
    uid = (C.c_char * 128)()
    call("mmvae_comm_unique_id", C.cast(uid, C.c_void_p))
    comm = C.c_void_p()
    call("mmvae_comm_init", C.byref(comm), 0, 1, C.cast(uid, C.c_void_p))
    try:
        assert call("mmvae_comm_world", comm) == 1
        g = torch.randn(1 << 20, device=dev)
        want = g.clone()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call("mmvae_allreduce_grads", comm, ptr(g), g.numel(), s)
        torch.cuda.synchronize()
        assert torch.equal(g, want)                       # the sum over one rank
        call("mmvae_allreduce_grads", comm, ptr(g), 0, s)  # empty message: a no-op, not an error
    finally:
        call("mmvae_comm_destroy", comm)


def test_coco_caption_kernel_forms_agree():
    """The forms of the COCO caption recurrences must compute the same step: decoder with 8 / 4 workgroups per row block
    (cluster form: slices of the hidden units per rank, all-gathers through global memory; at 8 the COMPOSED form, whose
    layer-0 input projection reads h1 through W_ih0x W_ho, against the three-exchange kernels) against one workgroup per block,
    encoder with resident against streamed W_hh.  Same inputs, same draws; what may differ is fp32 summation order (split
    reductions) and the step's own run-to-run noise from fp32 atomics upstream, both far below the gates."""
    from multimodal_vae_amd.core import CocoState, FusedCocoStep
    from multimodal_vae_amd.init import default_init_
    dev = _dev()
    B, T = 32, 102
    g = torch.Generator().manual_seed(5)
    image = torch.rand(B, 3, 32, 32, generator=g).to(dev)
    text = (0.4 * torch.randn(B, T, 300, generator=g)).to(dev)
    eps = torch.randn(3, B, D, generator=g).to(dev)
    keep = (torch.rand(T, 3 * B, 200, generator=g) > 0.1).to(torch.uint8).to(dev)
    st = CocoState(D, dev, steps=T); default_init_(st, 21)
    eng = FusedCocoStep(st, B, 0.4 * torch.randn(300, generator=g), seed=3)
    from multimodal_vae_amd._lib import call
    switches = ("coco_no_comb", "coco_no_comb_bwd", "coco_no_mse_fuse")       # A/B knobs (include/mmvae_hip.h: mmvae_debug_set)

    def run(cluster, streamed, *off):
        """off: knobs that take parts of the composed cluster-of-8 form out (W_comb = W_ih0x W_ho, two exchanges per step,
        the MSE fused into the forward kernel's output pass); the default at cluster == 8 has all of them in."""
        for k in switches:
            call("mmvae_debug_set", k.encode(), 1 if k in off else 0)
        call("mmvae_debug_set", b"coco_cluster", cluster)
        call("mmvae_debug_set", b"coco_enc_streamed", 1 if streamed else 0)
        rt = torch.zeros(3, B, T, 300, device=dev)
        out = eng.forward_backward(image, text, True, True, eps=eps, gru_keep=keep, recon_text=rt)
        torch.cuda.synchronize()
        return rt, st.grads.clone(), out.losses().cpu().numpy()

    try:
        r0, g0, l0 = run(0, True)
        for cluster, streamed, *off in ((0, False), (4, False), (8, False), (8, True), (8, False, "coco_no_comb"),
                                        (8, False, "coco_no_comb_bwd"), (8, False, "coco_no_mse_fuse")):
            r, gg, l = run(cluster, streamed, *off)
            np.testing.assert_allclose(l, l0, rtol=2e-4)
            assert float((r - r0).abs().max()) < 1e-2, (cluster, streamed, off)
            assert float((gg - g0).norm() / g0.norm()) < 1e-2, (cluster, streamed, off)
            for n, shape, o in st.table:        # every tensor, so that a slice of units left out cannot hide in the norm
                k = int(np.prod(shape))
                a, b = gg[o:o + k], g0[o:o + k]
                assert float((a - b).norm()) <= 5e-2 * float(b.norm()) + 1e-7, (cluster, streamed, off, n)
    finally:
        for k in switches:
            call("mmvae_debug_set", k.encode(), 0)
        call("mmvae_debug_set", b"coco_cluster", 8)
        call("mmvae_debug_set", b"coco_enc_streamed", 0)
