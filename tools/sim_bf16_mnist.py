"""CPU emulation: how much do bf16 GEMM operands / bf16 raw activations perturb the MNIST MMVAE gradients?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import mmvae_ref as R
D=20
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
class RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return bf(x)
    @staticmethod
    def backward(ctx, g): return g
rnd = RoundSTE.apply
def run(B, mode):
    P = R.formula_params("mnist", D, requires_grad=True)
    image, label = R.formula_inputs("mnist", B); image = image.reshape(B,784)
    eps=[]
    for k in range(3):
        torch.manual_seed(100+k); eps.append(torch.empty(B,D).normal_())
    if mode == "fp32":
        losses, outs = R.mnist_step_losses(P, image, label, True, eps)
    else:
        # monkeypatch F.linear to round operands (and outputs when mode == 'bf16r')
        orig = F.linear
        def lin(x, w, b=None):
            y = orig(rnd(x), rnd(w), b)
            if mode == "bf16r" and y.shape[1] not in (2*D, 784, 10*0+10) : y = rnd(y)
            return y
        F.linear = lin
        try:
            losses, outs = R.mnist_step_losses(P, image, label, True, eps)
        finally:
            F.linear = orig
    (losses[0]+losses[1]+losses[2]).backward()
    return [l.item() for l in losses], outs, {n: P[n].grad.clone() for n in P if P[n].grad is not None}
for B in (8,128):
    l0,o0,g0 = run(B,"fp32")
    for mode in ("bf16op","bf16r"):
        l1,o1,g1 = run(B,mode)
        print("B",B,mode,"loss rel", [abs(a-b)/abs(a) for a,b in zip(l0,l1)])
        print("   mu err", [(o0[k][2]-o1[k][2]).abs().max().item() for k in range(3)])
        for n in g0:
            if g0[n].norm()>1e-6: print("   %-30s %.3e"%(n, ((g0[n]-g1[n]).norm()/g0[n].norm()).item()))
