"""Shared gradient comparison of the GPU parity tests (no absolute slack).

Every gradient tensor must match the oracle's within ``tensor_tol`` relative L2 and the total norm within ``total_tol``;
the tolerances at the call sites are about twice the error measured on MI355X (run with MMVAE_TOL_REPORT=1 to print the
worst tensor of every check).  The only tensors excused from the relative test are gradients that are analytically zero
(a bias in front of a BatchNorm, a parameter only an absent pass touches): reference norm below 1e-6 of the total -- the
engine's value for those must itself be below 1e-4 of the total norm."""
import os

import torch

REPORT = os.environ.get("MMVAE_TOL_REPORT") is not None


def check_gradients(pairs, tensor_tol, total_tol=None, label="", zero_names=()):
    """pairs: iterable of (name, engine_grad, reference_grad or None) tensors of equal numel."""
    rows = []
    tot_ref = tot_eng = 0.0
    items = []
    for name, gh, gr in pairs:
        gh = gh.detach().reshape(-1).double().cpu()
        gr = torch.zeros_like(gh) if gr is None else gr.detach().reshape(-1).double().cpu()
        assert gh.numel() == gr.numel(), name
        items.append((name, gh, gr))
        tot_ref += float(gr.pow(2).sum())
        tot_eng += float(gh.pow(2).sum())
    tot_ref, tot_eng = tot_ref ** 0.5, tot_eng ** 0.5
    worst = (0.0, "")
    for name, gh, gr in items:
        nr = float(gr.norm())
        if name in zero_names or nr < 1e-6 * tot_ref:
            assert float(gh.norm()) <= 1e-4 * tot_ref, (label, name, float(gh.norm()), tot_ref)
            continue
        rel = float((gh - gr).norm()) / nr
        worst = max(worst, (rel, name))
        rows.append((rel, name))
    tot_rel = abs(tot_eng - tot_ref) / tot_ref
    if REPORT:
        print("TOL %s: worst tensor %.3e (%s), total %.3e [gates %.1e / %s]" % (label, worst[0], worst[1], tot_rel, tensor_tol, total_tol))
    for rel, name in rows:
        assert rel <= tensor_tol, (label, name, rel, tensor_tol)
    if total_tol is not None:
        assert tot_rel <= total_tol, (label, tot_rel, total_tol)
    return worst[0], tot_rel
