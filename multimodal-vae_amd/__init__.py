"""MI355X-native engine for the MMVAE ELBO training step of wenxuanliu/multimodal-vae.

Python face mirrors the reference's ``<ds>/model.py`` + ``loss_function`` (see ``multimnist``); all hot-path
arithmetic runs in hand-written HIP kernels behind the C-ABI of ``libmmvae_hip.so`` (``include/mmvae_hip.h``).
"""
from ._lib import MMVAEError, LIB_PATH, load as load_library  # noqa: F401

__all__ = ["MMVAEError", "LIB_PATH", "load_library"]
