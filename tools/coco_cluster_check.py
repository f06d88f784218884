"""Cluster form of the COCO caption decoder forward against the single-workgroup kernel in ONE process (the knob
is read per call): the reconstructed captions must agree to the last bit when everything upstream is identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import core
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
from bench import synthetic_batch_for, synthetic_sos
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
st = core.CocoState(100, dev); default_init_(st, 1234)
a, b = synthetic_batch_for("coco", B, 1234)
a, b = a.to(dev), b.to(dev)
eng = core.FusedCocoStep(st, B, synthetic_sos(), seed=7)
g = torch.Generator().manual_seed(3)
eps = torch.randn(3, B, 100, generator=g).to(dev)
keep = (torch.rand(102, 3 * B, 200, generator=g) > 0.1).to(torch.uint8).to(dev)

def run(cluster):
    call("mmvae_debug_set", b"coco_cluster", cluster)
    rt = torch.zeros(3, B, 102, 300, device=dev)
    out = eng.forward_backward(a, b, True, True, eps=eps, gru_keep=keep, recon_text=rt)
    torch.cuda.synchronize()
    return rt.clone(), st.grads.clone(), out.losses().cpu().numpy()

r0, g0, l0 = run(0)
r0b, g0b, _ = run(0)
print("baseline twice: recon equal", torch.equal(r0, r0b), "max diff", float((r0 - r0b).abs().max()), "grads rel", float((g0 - g0b).norm() / g0.norm()))
for P in (4, 8):
    r, gg, l = run(P)
    print("cluster", P, ": recon equal", torch.equal(r, r0), "max diff", float((r - r0).abs().max()), "grads rel", float((gg - g0).norm() / g0.norm()), l - l0)
for P in (0, 4, 8):
    call("mmvae_debug_set", b"coco_cluster", P)
    for _ in range(3): eng(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng(a, b)
    torch.cuda.synchronize(); print("cluster", P, "ms/step %.3f" % ((time.perf_counter() - t0) * 100))
