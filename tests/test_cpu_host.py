"""CPU-only checks: the C-ABI library loads and exports every symbol of include/mmvae_hip.h, the host-side plan
(parameter table, pack tables, workspace) agrees with the oracle's table, the Python face has the reference's
state_dict surface, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import mmvae_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from multimodal_vae_amd import build  # type: ignore  # noqa: F401
        import importlib.util
        spec = importlib.util.spec_from_file_location("mmvae_build", os.path.join(ROOT, "multimodal-vae_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); mod.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "mmvae_hip.h")).read()
    names = set(re.findall(r"\b(mmvae_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 40
    for n in sorted(names):
        assert hasattr(lib, n), "library does not export %s" % n
    from multimodal_vae_amd import _lib
    assert set(_lib.SIGNATURES) <= names | {"mmvae_last_error", "mmvae_version"}


@pytest.mark.parametrize("D", [20, 100])
def test_param_table_matches_reference_state_dict_order(lib, D):
    from multimodal_vae_amd._lib import call
    h = call("mmvae_mm_create", D, 8)
    assert h
    table = R.param_table("multimnist", D)
    assert call("mmvae_mm_num_params", h) == len(table) == 52
    name = C.create_string_buffer(128); nd = C.c_int(); off = C.c_longlong(); shape = (C.c_int * 4)()
    expect_off = 0
    for i, (n, s) in enumerate(table):
        call("mmvae_mm_param_info", h, i, name, C.byref(nd), shape, C.byref(off))
        assert name.value.decode() == n and tuple(shape[k] for k in range(nd.value)) == tuple(s) and off.value == expect_off
        expect_off += int(np.prod(s))
    assert call("mmvae_mm_param_count", h) == expect_off
    assert call("mmvae_mm_workspace_bytes", h) > 0
    assert call("mmvae_mm_packed_elems", h) > expect_off           # forward + backward packings
    assert call("mmvae_mm_num_bn", h) == 6 and call("mmvae_mm_bn_floats", h) == 2 * (64 + 128 + 256 + 128 + 64 + 32)
    call("mmvae_mm_destroy", h)


def test_error_reporting_without_exceptions(lib):
    from multimodal_vae_amd._lib import call, MMVAEError
    assert call("mmvae_mm_create", 0, 8) is None                   # invalid n_latents -> NULL + message
    assert b"n_latents" in lib.mmvae_last_error()
    h = call("mmvae_mm_create", 100, 4)
    with pytest.raises(MMVAEError):                                # unbound plan
        call("mmvae_mm_pack_weights", h, None)
    call("mmvae_mm_destroy", h)


def test_python_face_has_the_reference_state_dict():
    from multimodal_vae_amd import multimnist as M
    vae = M.MultimodalVAE(100)
    sd = vae.state_dict()
    P = R.formula_params("multimnist", 100)
    assert list(sd.keys()) == list(P.keys()) or set(sd.keys()) == set(P.keys())
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert [n for n, _ in vae.named_parameters()] == [n for n, _ in R.param_table("multimnist", 100)]
    assert M.elbo_loss is M.loss_function
    assert (M.max_length, M.n_characters, M.SOS, M.FILL) == (4, 12, 10, 11)      # multimnist/utils.py:14-19
    enc = M.ImageEncoder(100)                                                     # standalone modules keep local names
    assert "features.0.weight" in enc.state_dict() and "classifier.6.bias" in enc.state_dict()


def test_product_path_refuses_cpu():
    from multimodal_vae_amd import multimnist as M, MMVAEError
    vae = M.MultimodalVAE(100)
    with pytest.raises(MMVAEError):
        vae(image=torch.zeros(2, 1, 50, 50))
    with pytest.raises(MMVAEError):
        M.ProductOfExperts()(torch.zeros(2, 3, 4), torch.zeros(2, 3, 4))
    with pytest.raises(AssertionError):
        vae()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multimodal-vae_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no oracle", ""), fn


# ---------------------------------------------------------------------------------------------- MNIST (mnist/model.py)
def test_mnist_param_table_and_state_dict(lib):
    from multimodal_vae_amd._lib import call
    from multimodal_vae_amd import mnist as M
    D = 20
    h = call("mmvae_mnist_create", D, 128)
    assert h
    table = R.param_table("mnist", D)
    assert call("mmvae_mnist_num_params", h) == len(table) == 31
    name = C.create_string_buffer(128); nd = C.c_int(); off = C.c_longlong(); shape = (C.c_int * 4)()
    expect_off = 0
    for i, (n, s) in enumerate(table):
        call("mmvae_mnist_param_info", h, i, name, C.byref(nd), shape, C.byref(off))
        assert name.value.decode() == n and tuple(shape[k] for k in range(nd.value)) == tuple(s) and off.value == expect_off
        expect_off += int(np.prod(s))
    assert call("mmvae_mnist_param_count", h) == expect_off
    assert call("mmvae_mnist_num_bn", h) == 6 and call("mmvae_mnist_bn_floats", h) == 2 * (400 + 200 + 200 + 400 + 50 + 10)
    ch = C.c_int()
    call("mmvae_mnist_bn_info", h, 4, name, C.byref(ch), C.byref(off))
    assert name.value.decode() == "text_encoder.net.1" and ch.value == 50
    assert call("mmvae_mnist_workspace_bytes", h) > 0
    call("mmvae_mnist_destroy", h)
    assert call("mmvae_mnist_create", 0, 8) is None and b"n_latents" in lib.mmvae_last_error()
    vae = M.MultimodalVAE(D)
    sd = vae.state_dict()
    P = R.formula_params("mnist", D)
    assert set(sd.keys()) == set(P.keys())
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert [n for n, _ in vae.named_parameters()] == [n for n, _ in table]
    assert M.elbo_loss is M.loss_function
    with pytest.raises(M.MMVAEError):
        vae(image=torch.zeros(2, 784))
    with pytest.raises(AssertionError):
        vae()


def test_bf16_contract_is_off_by_default_and_scoped():
    """The oracle's bf16 storage-contract emulation must never leak into the fp32 reference restatement."""
    P = R.formula_params("mnist", 20)
    image, label = R.formula_inputs("mnist", 8)
    eps = [R.formula_eps(8, 20, k) for k in range(3)]
    with torch.no_grad():
        a, _ = R.mnist_step_losses({k: v.clone() for k, v in P.items()}, image, label, True, eps)
        with R.bf16_contract():
            b, _ = R.mnist_step_losses({k: v.clone() for k, v in P.items()}, image, label, True, eps)
        c, _ = R.mnist_step_losses({k: v.clone() for k, v in P.items()}, image, label, True, eps)
    assert [x.item() for x in a] == [x.item() for x in c]
    assert [x.item() for x in a] != [x.item() for x in b]
    for x, y in zip(a, b):
        assert abs(x.item() - y.item()) <= 1e-3 * abs(x.item())          # the contract moves the ELBO by < 1e-3 relative


# ---------------------------------------------------------------------------------------------- CelebA (celeba/model.py)
def test_celeba_param_table_and_state_dict(lib):
    from multimodal_vae_amd._lib import call
    from multimodal_vae_amd import celeba as M
    D = 100
    h = call("mmvae_celeba_create", D, 16)
    assert h
    table = R.param_table("celeba", D)
    assert call("mmvae_celeba_num_params", h) == len(table) == 38
    name = C.create_string_buffer(128); nd = C.c_int(); off = C.c_longlong(); shape = (C.c_int * 4)()
    expect_off = 0
    for i, (n, s) in enumerate(table):
        call("mmvae_celeba_param_info", h, i, name, C.byref(nd), shape, C.byref(off))
        assert name.value.decode() == n and tuple(shape[k] for k in range(nd.value)) == tuple(s) and off.value == expect_off
        expect_off += int(np.prod(s))
    assert call("mmvae_celeba_param_count", h) == expect_off == 8808802
    assert call("mmvae_celeba_num_bn", h) == 8
    assert call("mmvae_celeba_bn_floats", h) == 2 * (64 + 128 + 256 + 128 + 64 + 32 + 64 + 64)
    call("mmvae_celeba_destroy", h)
    assert call("mmvae_celeba_create", 99, 8) is None and b"n_latents" in lib.mmvae_last_error()
    vae = M.MultimodalVAE(D)
    sd = vae.state_dict()
    P = R.formula_params("celeba", D)
    assert set(sd.keys()) == set(P.keys())
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert [n for n, _ in vae.named_parameters()] == [n for n, _ in table]
    assert M.elbo_loss is M.loss_function and M.N_ATTRS == 18
    w0 = vae.image_encoder.features[0].weight.detach().clone()
    vae.weight_init(0.0, 0.02)                                   # celeba/model.py:24-26,121-123: touches nothing
    assert torch.equal(w0, vae.image_encoder.features[0].weight)
    with pytest.raises(M.MMVAEError):
        vae(image=torch.zeros(2, 3, 64, 64))
    with pytest.raises(AssertionError):
        vae()


def test_bench_reference_flop_constants_match_the_counted_fixture():
    """bench.py prices the step with the reference's FLOP count; the numbers are counted on the imported reference by
    `python oracle/make_golden.py --flops` (FlopCounterMode over the 3 passes + backward of multimnist/train.py:154-168 and its
    CelebA / COCO twins) and committed as tests/golden/reference_flops.json."""
    import json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    t = json.load(open(os.path.join(root, "tests", "golden", "reference_flops.json")))
    for ds in ("multimnist", "celeba", "coco"):
        assert abs(bench.FLOP_PER_SAMPLE[ds] - t[ds]["flops_per_sample_fwd_bwd"]) / t[ds]["flops_per_sample_fwd_bwd"] < 1e-4, ds
