import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
sys.path.insert(0, '.')
from bench import synthetic_batch
dev = torch.device('cuda:0'); B = 256
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234); img, txt = img.to(dev), txt.to(dev)
eng = FusedELBOStep(st, B)
for _ in range(10): eng(img, txt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    eng(img, txt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue per step (ms):', (t1 - t0) / 50 * 1e3, ' total per step (ms):', (t2 - t0) / 50 * 1e3)
