import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodal_vae_amd
from multimodal_vae_amd.core import MnistState, FusedMnistStep
from oracle import mmvae_ref as R
D = 20
dev = torch.device("cuda:0")
for B in (8, 128):
    P = R.formula_params("mnist", D, requires_grad=True)
    st = MnistState(D, dev)
    for n, shape, off in st.table:
        st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(dev)
    image, label = R.formula_inputs("mnist", B)
    image = image.reshape(B, 784)
    eps = []
    for k in range(3):
        torch.manual_seed(100 + k); eps.append(torch.empty(B, D).normal_())
    eng = FusedMnistStep(st, B)
    mu = torch.zeros(3, B, D, device=dev); lv = torch.zeros(3, B, D, device=dev)
    out = eng.forward_backward(image.to(dev).contiguous(), label.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous(), mu=mu, logvar=lv)
    o_losses, o_outs = R.mnist_step_losses(P, image, label, True, eps)
    (o_losses[0] + o_losses[1] + o_losses[2]).backward()
    print("B", B, "loss", out.losses().cpu().numpy(), [l.item() for l in o_losses])
    for k in range(3):
        print(" mu err", (mu[k].cpu() - o_outs[k][2]).abs().max().item(), "lv err", (lv[k].cpu() - o_outs[k][3]).abs().max().item())
    g = st.grads.cpu()
    for n, shape, off in st.table:
        gr = P[n].grad.reshape(-1); gh = g[off:off + gr.numel()]
        print("  %-32s ref %.4e got %.4e relerr %.3e" % (n, gr.norm().item(), gh.norm().item(), (gh - gr).norm().item() / max(gr.norm().item(), 1e-12)))
