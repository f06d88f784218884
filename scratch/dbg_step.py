import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from oracle import mmvae_ref as R
torch.manual_seed(0)
dev = torch.device('cuda:0')
B, D = 8, 100
fx = np.load('tests/golden/multimnist_b8.npz')
P = R.formula_params('multimnist', D, requires_grad=True)
st = MultimnistState(D, dev)
names = [n for n, _ in R.param_table('multimnist', D)]
assert [t[0] for t in st.table] == names, "param table mismatch"
for n, shape, off in st.table:
    assert tuple(P[n].shape) == tuple(shape), (n, shape, P[n].shape)
    st.params[off:off + P[n].numel()] = P[n].detach().reshape(-1).to(dev)
image, text = R.formula_inputs('multimnist', B)
eps = [torch.from_numpy(fx[f'eps_{k}']) for k in range(3)]
ft = torch.from_numpy(np.stack([fx[f'tokens_{k}'] for k in range(3)])).long()   # [3][B][4]
eng = FusedELBOStep(st, B)
eng.enc_dropout = eng.gru_dropout = False
mu = torch.zeros(3, B, D, device=dev); lv = torch.zeros(3, B, D, device=dev)
ri = torch.zeros(3, B, 2500, device=dev); rt = torch.zeros(3, B, 4, 12, device=dev); tk = torch.zeros(3, B, 4, dtype=torch.int64, device=dev)
out = eng.forward_backward(image.to(dev), text.to(dev), True, True, eps=torch.stack(eps).to(dev).contiguous(),
                           force_tokens=ft.reshape(3 * B, 4).to(dev).contiguous(), recon_image=ri, recon_text=rt, mu=mu, logvar=lv, tokens=tk)
torch.cuda.synchronize()
print('losses hip', out.losses().cpu().numpy(), 'ref', fx['loss'])
# oracle
losses, outs = R.multimnist_step_losses(P, image, text, True, 1e-3, eps, None, None, [ft[0], ft[1], ft[2]], 0.0, 0.0)
(losses[0] + losses[1] + losses[2]).backward()
for k in range(3):
    print(f'pass{k} mu err', (mu[k].cpu() - outs[k][2]).abs().max().item(), 'lv err', (lv[k].cpu() - outs[k][3]).abs().max().item(),
          'recon err', (ri[k].cpu() - outs[k][0].reshape(B, 2500)).abs().max().item(),
          'words err', (rt[k].cpu() - outs[k][1]).abs().max().item(), 'tok agree', (tk[k].cpu() == outs[k][4]).float().mean().item())
g = st.grads.cpu()
tot_ref = torch.sqrt(sum(P[n].grad.double().pow(2).sum() for n in names)).item()
worst = []
for n, shape, off in st.table:
    gr = P[n].grad.reshape(-1); gh = g[off:off + gr.numel()]
    rel = (gh - gr).norm().item() / max(gr.norm().item(), 1e-6 * tot_ref)
    worst.append((rel, n, gr.norm().item(), gh.norm().item()))
for rel, n, a, b in worst:
    print(f'{n:45s} rel_l2_err {rel:9.3e} |ref| {a:9.3e} |hip| {b:9.3e}')
print('total grad norm hip', g.double().norm().item(), 'ref', tot_ref)

# ---- layer-by-layer decoder check
import torch.nn.functional as F
from multimodal_vae_amd._lib import call
def wsbuf(name, dtype, shape):
    o = call("mmvae_mm_debug_offset", eng.h, name.encode())
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    return eng.ws[o:o + n].view(dtype).view(shape).float().cpu()
with torch.no_grad():
    Pd = {k: v.detach() for k, v in P.items()}
    Pd2 = R.formula_params('multimnist', D)
    z = torch.cat([eps[k] * torch.exp(0.5 * outs[k][3].detach()) + outs[k][2].detach() for k in range(3)])
    zh = wsbuf('z_f32', torch.float32, (3 * B, D))
    print('z err', (zh - z).abs().max().item())
    u = F.linear(z, Pd['image_decoder.upsample.0.weight'], Pd['image_decoder.upsample.0.bias'])
    uh = wsbuf('u', torch.bfloat16, (3 * B, 2, 2, 256)).permute(0, 3, 1, 2).reshape(3 * B, 1024)
    print('u err', (uh - u).abs().max().item(), u.abs().max().item())
    x = R.swish(u).view(-1, 256, 2, 2)
    def bn_groups(x, pre):
        outl = []
        for g in range(3):
            xg = x[g * B:(g + 1) * B]
            m = xg.mean((0, 2, 3), keepdim=True); v = xg.var((0, 2, 3), unbiased=False, keepdim=True)
            outl.append((xg - m) / torch.sqrt(v + 1e-5) * Pd[pre + '.weight'].view(1, -1, 1, 1) + Pd[pre + '.bias'].view(1, -1, 1, 1))
        return torch.cat(outl)
    q1 = F.conv_transpose2d(x, Pd['image_decoder.hallucinate.0.weight'], None, 2, 0)
    q1h = wsbuf('q1', torch.bfloat16, (3 * B, 6, 6, 128)).permute(0, 3, 1, 2)
    e = (q1h - q1).abs(); print('q1 err', e.max().item(), q1.abs().max().item(), 'argmax', np.unravel_index(e.argmax().item(), e.shape))
    x = R.swish(bn_groups(q1, 'image_decoder.hallucinate.1'))
    q2 = F.conv_transpose2d(x, Pd['image_decoder.hallucinate.3.weight'], None, 2, 1)
    q2h = wsbuf('q2', torch.bfloat16, (3 * B, 12, 12, 64)).permute(0, 3, 1, 2)
    e = (q2h - q2).abs(); print('q2 err', e.max().item(), q2.abs().max().item(), 'argmax', np.unravel_index(e.argmax().item(), e.shape))
    x = R.swish(bn_groups(q2, 'image_decoder.hallucinate.4'))
    q3 = F.conv_transpose2d(x, Pd['image_decoder.hallucinate.6.weight'], None, 2, 1)
    q3h = wsbuf('q3', torch.bfloat16, (3 * B, 25, 25, 32)).permute(0, 3, 1, 2)
    e = (q3h - q3).abs(); print('q3 err', e.max().item(), q3.abs().max().item(), 'argmax', np.unravel_index(e.argmax().item(), e.shape))
    x = R.swish(bn_groups(q3, 'image_decoder.hallucinate.7'))
    lg = F.conv_transpose2d(x, Pd['image_decoder.hallucinate.9.weight'], None, 2, 1)
    lgh = lg
    e = (lgh - lg).abs(); print('logit err', e.max().item(), lg.abs().max().item(), 'argmax', np.unravel_index(e.argmax().item(), e.shape))
    print('mean abs logit err', e.mean().item())
    e = (q3h - q3).abs()
    for ph in range(2):
        for pw in range(2):
            ee = e[:, :, ph::2, pw::2]
            print('q3 class', ph, pw, 'max', ee.max().item(), 'mean', ee.mean().item())
    print('per-n max', e.amax((1,2,3))[:24])
    print('per-y max', e.amax((0,1,3)))
