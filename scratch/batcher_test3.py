import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd import data as D
from multimodal_vae_amd.utils import charlist_tensor
from multimodal_vae_amd._lib import call, ptr
dev = torch.device("cuda", 0)
B = 256
x, y = D.synthetic_multimnist(B * 40, seed=1)
t = torch.stack([charlist_tensor(l) for l in y])
L = D.DeviceBatcher(x, t, B, dev, shuffle=True, seed=0)
state = MultimnistState(100, dev); default_init_(state, seed=1234)
eng = FusedELBOStep(state, B, lr=1e-3, seed=1234)
for _ in range(20): eng(L.dev_f32[0], L.dev_tx[0])
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
order = torch.randperm(len(x))
nb = len(L)
def stage(slot, idx):
    t0 = time.perf_counter(); L.consumed[slot].synchronize(); tick("consumed.sync", t0)
    t0 = time.perf_counter(); torch.index_select(L.images, 0, idx, out=L.stage_u8[slot]); torch.index_select(L.text, 0, idx, out=L.stage_tx[slot]); tick("gather", t0)
    t0 = time.perf_counter()
    with torch.cuda.stream(L.copy_stream):
        L.dev_u8[slot].copy_(L.stage_u8[slot], non_blocking=True)
        L.dev_tx[slot].copy_(L.stage_tx[slot], non_blocking=True)
        L.ready[slot].record(L.copy_stream)
    tick("h2d enqueue", t0)
T0 = time.perf_counter()
stage(0, order[0:B])
for b in range(nb):
    slot = b & 1
    if b + 1 < nb: stage(slot ^ 1, order[(b + 1) * B:(b + 2) * B])
    cur = torch.cuda.current_stream(dev)
    t0 = time.perf_counter(); cur.wait_event(L.ready[slot]); tick("wait_event", t0)
    t0 = time.perf_counter(); call("mmvae_u8_to_f32", ptr(L.dev_u8[slot]), L.dev_u8[slot].numel(), 255.0, ptr(L.dev_f32[slot]), C.c_void_p(cur.cuda_stream)); tick("u8_to_f32", t0)
    t0 = time.perf_counter(); eng(L.dev_f32[slot], L.dev_tx[slot]); tick("step enqueue", t0)
    t0 = time.perf_counter(); L.consumed[slot].record(cur); tick("record", t0)
torch.cuda.synchronize()
print("total ms/step %.3f" % ((time.perf_counter() - T0) / nb * 1e3))
for k, v in acc.items(): print("  %-16s %.3f ms/step" % (k, v / nb * 1e3))
