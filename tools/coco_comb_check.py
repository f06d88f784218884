"""Composed (two-exchange) caption decoder kernels against the three-exchange cluster kernels in ONE process: per-tensor
gradient differences (the knobs are read per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import core
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
from bench import synthetic_batch_for, synthetic_sos
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
st = core.CocoState(100, dev); default_init_(st, 1234)
a, b = synthetic_batch_for("coco", B, 1234)
a, b = a.to(dev), b.to(dev)
eng = core.FusedCocoStep(st, B, synthetic_sos(), seed=7)
g = torch.Generator().manual_seed(3)
eps = torch.randn(3, B, 100, generator=g).to(dev)
keep = (torch.rand(102, 3 * B, 200, generator=g) > 0.1).to(torch.uint8).to(dev)

def run(env):
    for k in ("coco_no_comb", "coco_no_comb_bwd"):
        call("mmvae_debug_set", k.encode(), 1 if k in env else 0)
    rt = torch.zeros(3, B, 102, 300, device=dev)
    out = eng.forward_backward(a, b, True, True, eps=eps, gru_keep=keep, recon_text=rt)
    torch.cuda.synchronize()
    return rt.clone(), st.grads.clone(), out.losses().cpu().numpy()

r0, g0, l0 = run({"coco_no_comb"})
for name, env in (("fwd composed", {"coco_no_comb_bwd"}), ("fwd+bwd composed", set())):
    r, gg, l = run(env)
    print(name, ": recon max diff", float((r - r0).abs().max()), "grads rel", float((gg - g0).norm() / g0.norm()), l - l0)
    for n, shape, off in st.table:
        k = int(np.prod(shape))
        x, y = gg[off:off + k], g0[off:off + k]
        rel = float((x - y).norm() / (y.norm() + 1e-30))
        if rel > 5e-3:
            print("   %-40s rel %.3e  norm %.3e" % (n, rel, float(y.norm())))
