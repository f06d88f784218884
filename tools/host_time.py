"""Host enqueue time of the fused step against its GPU time (is the step host-bound?).  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd  # noqa
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from bench import synthetic_batch
dev = torch.device('cuda:0'); B = 256
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234)
img, txt = img.to(dev), txt.to(dev)
eng = FusedELBOStep(st, B)
for _ in range(20): eng(img, txt)
torch.cuda.synchronize()
import gc; gc.disable()
N = 300
t0 = time.perf_counter()
for _ in range(N): eng(img, txt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/step   total %.1f us/step   (GPU drained %.1f us after the last enqueue)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, (t2 - t1) * 1e6))
# host alone: enqueue while the GPU is idle behind a long sleep kernel?  approximate by timing enqueue of one step after sync
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    a = time.perf_counter(); eng(img, txt); b = time.perf_counter()
    ts.append((b - a) * 1e6)
ts.sort()
print("enqueue of one step on an idle GPU: median %.1f us  min %.1f us" % (ts[len(ts) // 2], ts[0]))
