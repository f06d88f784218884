import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodal_vae_amd
from multimodal_vae_amd.core import CelebaState, FusedCelebaStep
from oracle import mmvae_ref as R
D = 100
dev = torch.device("cuda:0")
Bs = [int(a) for a in sys.argv[1:]] or [4]
for B in Bs:
    P = R.formula_params("celeba", D, requires_grad=True)
    st = CelebaState(D, dev)
    for n, shape, off in st.table:
        st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(dev)
    image, attrs = R.formula_inputs("celeba", B)
    eps = []
    for k in range(3):
        torch.manual_seed(100 + k); eps.append(torch.empty(B, D).normal_())
    eng = FusedCelebaStep(st, B); eng.enc_dropout = False
    mu = torch.zeros(3, B, D, device=dev); lv = torch.zeros(3, B, D, device=dev)
    ra = torch.zeros(3, B, 18, device=dev); ri = torch.zeros(3, B, 3, 64, 64, device=dev)
    out = eng.forward_backward(image.to(dev).contiguous(), attrs.to(dev).contiguous(), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                               mu=mu, logvar=lv, recon_attrs=ra, recon_image=ri)
    torch.cuda.synchronize()
    o_losses, o_outs = R.celeba_step_losses(P, image, attrs, True, eps, None, 0.0)
    (o_losses[0] + o_losses[1] + o_losses[2]).backward()
    print("B", B, "loss", out.losses().cpu().numpy(), [l.item() for l in o_losses])
    print(" parts", [p.cpu().numpy() for p in out.parts()])
    for k in range(3):
        print(" mu err", (mu[k].cpu() - o_outs[k][2]).abs().max().item(), "lv err", (lv[k].cpu() - o_outs[k][3]).abs().max().item(),
              "ri err", (ri[k].cpu() - o_outs[k][0]).abs().max().item(), "ra err", (ra[k].cpu() - o_outs[k][1]).abs().max().item())
    g = st.grads.cpu()
    for n, shape, off in st.table:
        gr = P[n].grad.reshape(-1); gh = g[off:off + gr.numel()]
        print("  %-36s ref %.4e got %.4e relerr %.3e" % (n, gr.norm().item(), gh.norm().item(), (gh - gr).norm().item() / max(gr.norm().item(), 1e-12)))
