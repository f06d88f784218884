"""Which part of the loader slows the loader-fed step down?  Knock-outs of data.DeviceBatcher's per-batch actions (results are
garbage, only the clock matters).  Run on the GPU box."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import core, data
from multimodal_vae_amd.data import DeviceBatcher
from multimodal_vae_amd.init import default_init_
from bench import synthetic_batch_for
dev = torch.device("cuda:0"); B = 256
a, b = synthetic_batch_for("multimnist", 8 * B, 7)
u8 = (a * 255).round().to(torch.uint8)[:, 0]
st = core.MultimnistState(100, dev); default_init_(st, 1); eng = core.FusedELBOStep(st, B)
import gc; gc.disable()

def run(L, n, step=True):
    done = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while done < n:
        for im, tx in L:
            if step: eng(im, tx)
            done += 1
            if done == n: break
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3

ims, txs = a[:B].to(dev).contiguous(), b[:B].to(dev).contiguous()
for _ in range(50): eng(ims, txs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): eng(ims, txs)
torch.cuda.synchronize(); print(f"resident                         wall {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms")
for hp in (0, 1):
    for S in (3, 4, 6, 8):
        os.environ["MMVAE_LOADER_HOST_PACED"] = str(hp)
        L = DeviceBatcher(u8, b, B, dev, shuffle=True, seed=1, slots=S)
        run(L, 30)
        print("full loader, %2d slots, host paced %d   host %.3f wall %.3f" % ((S, hp) + run(L, 300)))
os.environ["MMVAE_LOADER_HOST_PACED"] = "0"
L = DeviceBatcher(u8, b, B, dev, shuffle=True, seed=1, slots=3)
run(L, 30)
orig_copy, orig_gather = L._copy, L._gather
L._copy = lambda slot: None
print("no H2D copies / events           host %.3f wall %.3f" % run(L, 300))
L._copy = orig_copy
L._gather = lambda slot, ix: None
print("no host gather                   host %.3f wall %.3f" % run(L, 300))
L._gather = orig_gather
real_call = data.call
data.call = lambda name, *args: None if name == "mmvae_u8_to_f32" else real_call(name, *args)
print("no u8->f32 kernel                host %.3f wall %.3f" % run(L, 300))
data.call = real_call
class NoEv:
    def record(self, *a): pass
    def synchronize(self): pass
    def wait(self, *a): pass
ev_r, ev_c = L.ready, L.consumed
L.consumed = [NoEv() for _ in ev_c]
print("no consumed events               host %.3f wall %.3f" % run(L, 300))
L.consumed = ev_c
