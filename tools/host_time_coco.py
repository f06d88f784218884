"""Host enqueue time of the fused COCO step against its GPU time (is the step host-bound?).  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import core
from multimodal_vae_amd.init import default_init_
from bench import synthetic_batch_for, synthetic_sos
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda:0')
st = core.CocoState(100, dev); default_init_(st, 1234)
a, b = synthetic_batch_for("coco", B, 1234)
a, b = a.to(dev), b.to(dev)
eng = core.FusedCocoStep(st, B, synthetic_sos(), seed=7)
for _ in range(10): eng(a, b)
torch.cuda.synchronize()
import gc; gc.disable()
N = 100
t0 = time.perf_counter()
for _ in range(N): eng(a, b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/step   total %.1f us/step   (GPU drained %.1f us after the last enqueue)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, (t2 - t1) * 1e6))
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    x = time.perf_counter(); eng(a, b); y = time.perf_counter()
    ts.append((y - x) * 1e6)
ts.sort()
print("enqueue of one step on an idle GPU: median %.1f us  min %.1f us" % (ts[len(ts) // 2], ts[0]))
