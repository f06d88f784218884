"""Ring-staged weight-gradient kernel (csrc/wgrad_ring.hip) against the streamed kernel (gemm.hip wgrad_kernel) on the activations
and gradients of a real MultiMNIST step: the conv / transposed-conv layers of multimnist/model.py:160-169,199-208, same operands,
the packed fp32 gradients compared element by element -- with partial copies + reduce launch and with the fp32-atomic epilogue."""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

LAYERS = ["dec_convT3_wgrad", "dec_convT2_wgrad", "dec_convT1_wgrad", "enc_conv2_wgrad", "enc_conv3_wgrad", "enc_conv4_wgrad"]


@pytest.mark.parametrize("B", [16, 256])
def test_ring_kernel_equals_streamed_kernel(B):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd._lib import call
    from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
    from multimodal_vae_amd.init import default_init_
    from bench import synthetic_batch
    dev = torch.device("cuda:0")
    st = MultimnistState(100, dev); default_init_(st, 1234)
    img, txt = synthetic_batch(B, 1234)
    eng = FusedELBOStep(st, B)
    eng(img.to(dev), txt.to(dev)); st.ensure_packed(); torch.cuda.synchronize()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(layer, ring, atomic_kb, pair=0):
        call("mmvae_debug_set", b"wgrad_ring", ring)
        call("mmvae_debug_set", b"wr_pair", pair)
        call("mmvae_debug_set", b"wr_atomic_kb", atomic_kb)
        st.gpk.zero_()
        call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer.encode(), 1, sp)
        torch.cuda.synchronize()
        return st.gpk.clone()
    try:
        for layer in LAYERS:
            ref = run(layer, 0, 256)
            assert int((ref != 0).sum()) > 0 and bool(torch.isfinite(ref).all()), layer
            for atomic_kb in (0, 4096):                      # partial copies + reduce / fp32 atomics into the packed gradient
                for pair in (1, 0):                          # two parity classes per 8-wave workgroup / one per 4-wave workgroup
                    got = run(layer, 1, atomic_kb, pair)
                    rel = float((ref - got).norm() / ref.norm())
                    assert rel < 2e-5, (layer, atomic_kb, pair, rel)   # same bf16 operands, fp32 accumulation in another order
                    assert torch.equal(got == 0, ref == 0) or rel < 2e-5
    finally:
        call("mmvae_debug_set", b"wgrad_ring", 1)
        call("mmvae_debug_set", b"wr_atomic_kb", 256)
        call("mmvae_debug_set", b"wr_pair", 0)
