"""Weak-supervision step variants of the reference: per batch, a random subset of the three passes.

* ``paired_weak`` (multimnist/paired_weak.py:84-117, mnist/paired_weak.py): with probability ``weak_perc`` the batch is
  a PAIRED example -> all three passes, the image-only pass with ``lambda_yx = 1`` (not 0.5); otherwise only the two
  uni-modal passes, each scored on its own modality alone (``recon_text=None`` / ``recon_image=None`` in the reference
  = weight 0 here).
* ``modal_weak`` (multimnist/modal_weak.py:87-117, mnist/modal_weak.py): the joint pass always; the image-only pass
  with probability ``weak_perc_m1`` (weights 1, 1), the text-only pass with probability ``weak_perc_m2`` (weights 0, 1).

Each function returns ``dict(passes=..., lambda_xy=..., lambda_yx=...)`` for ``FusedTrainer.__call__`` /
``FusedELBOStep.forward_backward``; randomness comes from ``numpy.random.random()`` exactly like the reference.
"""
import numpy as np


def paired_weak(weak_perc: float, rng=np.random) -> dict:
    if rng.random() < weak_perc:
        return dict(passes=(True, True, True), lambda_xy=(1.0, 1.0, 0.0), lambda_yx=(1.0, 1.0, 1.0))
    return dict(passes=(False, True, True), lambda_xy=(0.0, 1.0, 0.0), lambda_yx=(0.0, 0.0, 1.0))


def modal_weak(weak_perc_m1: float, weak_perc_m2: float, rng=np.random) -> dict:
    show_image = rng.random() < weak_perc_m1
    show_text = rng.random() < weak_perc_m2
    return dict(passes=(True, bool(show_image), bool(show_text)), lambda_xy=(1.0, 1.0, 0.0), lambda_yx=(1.0, 1.0, 1.0))
