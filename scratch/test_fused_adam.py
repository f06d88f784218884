import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
import bench
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep, _FusedStepBase
from multimodal_vae_amd.init import default_init_
dev = torch.device("cuda:0")
B = 64
img, txt = bench.synthetic_batch(B, 3)
img, txt = img.to(dev), txt.to(dev)
res = []
for fused in (True, False, False):
    st = MultimnistState(100, dev); default_init_(st, 7)
    eng = FusedELBOStep(st, B, seed=11)
    eng.enc_dropout = eng.gru_dropout = False
    eps = torch.randn(3, B, 100, generator=torch.Generator().manual_seed(1)).to(dev)
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
        if fused: eng(img, txt, eps=eps)
        else: _FusedStepBase.__call__(eng, img, txt, eps=eps)
    torch.cuda.synchronize()
    res.append((st.params.clone(), st.grads.clone(), eng.exp_avg.clone()))
for (i, j, tag) in ((0, 1, "fused vs unfused"), (1, 2, "unfused vs unfused")):
    for a, b, n in zip(res[i], res[j], ("params", "grads", "exp_avg")):
        print(tag, n, "max abs diff", (a - b).abs().max().item(), "rel", ((a - b).norm() / b.norm()).item())
m = st.grad_map()
print("mapped", int((m >= 0).sum()), "vec", int((m < -1).sum()), "none", int((m == -1).sum()), "of", st.nparams)
