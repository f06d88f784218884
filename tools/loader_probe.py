"""Where does the loader's time go?  (run on the GPU box)  usage: loader_probe.py [multimnist|coco]"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import core
from multimodal_vae_amd.data import DeviceBatcher
from multimodal_vae_amd.init import default_init_
from bench import synthetic_batch_for, synthetic_sos
wl = sys.argv[1] if len(sys.argv) > 1 else "multimnist"
dev = torch.device("cuda:0")
B = 256 if wl == "multimnist" else 128
a, b = synthetic_batch_for(wl, 8 * B, 7)
u8 = (a * 255).round().to(torch.uint8)
if wl == "multimnist":
    u8 = u8[:, 0]
    st = core.MultimnistState(100, dev); default_init_(st, 1); eng = core.FusedELBOStep(st, B)
else:
    st = core.CocoState(100, dev); default_init_(st, 1); eng = core.FusedCocoStep(st, B, synthetic_sos())
import gc; gc.disable()
ims, txs = a[:B].to(dev).contiguous(), b[:B].to(dev).contiguous()
def resident(tag):
    for _ in range(30): eng(ims, txs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): eng(ims, txs)
    th = time.perf_counter() - t0; torch.cuda.synchronize()
    print(f"{wl}: resident step [{tag}] host {th / 200 * 1e3:.3f} ms  wall {(time.perf_counter() - t0) / 200 * 1e3:.3f}", flush=True)
resident("before any loader")
x = torch.empty(1 << 20).pin_memory(); resident("after a pinned allocation")
from multimodal_vae_amd._lib import OwnedStream
os_ = OwnedStream(dev); resident("after one more stream")
ev = torch.cuda.Event(); ev.record(); resident("after an event")
with torch.cuda.stream(os_.stream):
    y = x.to(dev, non_blocking=True)
torch.cuda.synchronize(); resident("after an async H2D copy on that stream")
for pin, cow in ((False, True),):
    try:
        L = DeviceBatcher(u8, b, B, dev, shuffle=True, seed=1, pin_dataset=pin, copy_on_worker=cow)
    except AssertionError as e:
        print("pin", pin, "not applicable:", e); continue
    def run(n, step, hold=None):
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
    run(20, True)
    print(f"{wl} pin={pin} copy_on_worker={cow}: loader alone  host {run(200, False)[0]:.3f} ms/iter  wall {run(200, False)[1]:.3f}")
    print(f"{wl} pin={pin} copy_on_worker={cow}: loader + step host {run(200, True)[0]:.3f} ms/iter  wall {run(200, True)[1]:.3f}")
ims, txs = a[:B].to(dev).contiguous(), b[:B].to(dev).contiguous()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): eng(ims, txs)
th = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"{wl}: resident step host {th / 200 * 1e3:.3f} ms  wall {(time.perf_counter() - t0) / 200 * 1e3:.3f}")
