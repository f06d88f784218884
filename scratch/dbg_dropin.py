import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd
from multimodal_vae_amd import multimnist as M
from oracle import mmvae_ref as R
dev = torch.device('cuda:0')
B, D = 8, 100
fx = np.load('tests/golden/multimnist_b8.npz')
P = R.formula_params('multimnist', D, requires_grad=True)
vae = M.MultimodalVAE(D, use_cuda=True)
sd = {k: v.detach().clone() for k, v in P.items()}
print(vae.load_state_dict(sd, strict=True))
vae.cuda(); vae.train()
vae.image_encoder.classifier[2].p = 0.0; vae.image_encoder.classifier[5].p = 0.0; vae.text_decoder.gru.dropout = 0.0
image, text = R.formula_inputs('multimnist', B)
imd, txd = image.to(dev), text.to(dev)
eps = [torch.from_numpy(fx[f'eps_{k}']) for k in range(3)]
ft = [torch.from_numpy(fx[f'tokens_{k}']).long() for k in range(3)]
opt = torch.optim.Adam(vae.parameters(), lr=1e-3)
opt.zero_grad()
outs = []
args = ((imd, txd), (imd, None), (None, txd))
lam = ((1., 1.), (1., .5), (0., 1.))
total = 0
for k in range(3):
    ri, rt, mu, lv = vae(image=args[k][0], text=args[k][1], eps=eps[k].to(dev), force_tokens=ft[k].to(dev))
    l = M.loss_function(mu, lv, recon_image=ri, image=imd, recon_text=rt, text=txd, kl_lambda=1e-3, lambda_xy=lam[k][0], lambda_yx=lam[k][1])
    print('loss', k, l.item(), fx['loss'][k])
    total = total + l
total.backward()
losses, o = R.multimnist_step_losses(P, image, text, True, 1e-3, eps, None, None, ft, 0.0, 0.0)
(losses[0] + losses[1] + losses[2]).backward()
worst = 0
for n, p in vae.named_parameters():
    gr = P[n].grad; gh = p.grad.cpu()
    rel = (gh - gr).norm().item() / max(gr.norm().item(), 1e-9)
    worst = max(worst, rel)
    if rel > 0.03: print(n, rel, gr.norm().item(), gh.norm().item())
print('worst rel', worst)
for n, b in vae.named_buffers():
    if n in fx.files or ('buf:' + n) in fx.files:
        print(n, (b.cpu() - torch.from_numpy(fx['buf:' + n])).abs().max().item())
opt.step()
# second forward after optimizer step (repack path)
ri, rt, mu, lv = vae(image=imd, text=txd)
print('after step ok', ri.shape, rt.shape, float(mu.abs().mean()))
vae.eval()
ri, rt, mu, lv = vae(image=imd, text=txd)
Pe = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
with torch.no_grad():
    o = R.multimnist_forward(Pe, image, text, False)
print('eval recon err', (ri.cpu() - o[0]).abs().max().item(), 'mu err', (mu.cpu() - o[2]).abs().max().item(), 'words err', (rt.cpu()-o[1]).abs().max().item())
