import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd import data as D
from multimodal_vae_amd.utils import charlist_tensor
dev = torch.device("cuda", 0)
B = 256
x, y = D.synthetic_multimnist(B * 40, seed=1)
t = torch.stack([charlist_tensor(l) for l in y])
loader = D.DeviceBatcher(x, t, B, dev, shuffle=True, seed=0)
def timeit(name, fn, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(name, "ms %.4f" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
# 1. loader alone
for ep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for im, tx in loader: n += 1
    torch.cuda.synchronize(); print("loader alone ms/batch %.4f" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
idx = torch.randperm(len(x))[:B]
timeit("index_select images", lambda: torch.index_select(x, 0, idx, out=loader.stage_u8[0]))
timeit("index_select text", lambda: torch.index_select(t, 0, idx, out=loader.stage_tx[0]))
timeit("h2d copy (current stream)", lambda: loader.dev_u8[0].copy_(loader.stage_u8[0], non_blocking=True))
def cp():
    with torch.cuda.stream(loader.copy_stream):
        loader.dev_u8[0].copy_(loader.stage_u8[0], non_blocking=True)
timeit("h2d copy (copy stream)", cp)
ev = torch.cuda.Event()
def evs():
    ev.record(); ev.synchronize()
timeit("event record+sync", evs)
state = MultimnistState(100, dev); default_init_(state, seed=1234)
eng = FusedELBOStep(state, B, lr=1e-3, seed=1234)
im, tx = loader.dev_f32[0], loader.dev_tx[0]
timeit("step resident", lambda: eng(im, tx), 100)
def step_copy_same():
    loader.dev_u8[0].copy_(loader.stage_u8[0], non_blocking=True)
    eng(im, tx)
timeit("step + h2d on the same stream", step_copy_same, 100)
def step_copy_other():
    with torch.cuda.stream(loader.copy_stream):
        loader.dev_u8[1].copy_(loader.stage_u8[1], non_blocking=True)
    eng(im, tx)
timeit("step + h2d on the copy stream", step_copy_other, 100)
