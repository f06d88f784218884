"""Where does the enqueue thread of a loader-fed MultiMNIST loop spend its time?  Per step: time inside the loader's iterator (waiting for
the worker's staged batch, the wait + ToTensor call, the consumed-event record) against time inside the training call (49 launches),
and how far the GPU lags the enqueue thread.  Run on the GPU box."""
import os, sys, time, torch, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import core
from multimodal_vae_amd.data import DeviceBatcher
from multimodal_vae_amd.init import default_init_
from bench import synthetic_batch_for
dev = torch.device("cuda:0"); B = 256
a, b = synthetic_batch_for("multimnist", 8 * B, 7)
u8 = (a * 255).round().clamp(0, 255).to(torch.uint8)[:, 0]
st = core.MultimnistState(100, dev); default_init_(st, 1)
eng = core.FusedELBOStep(st, B)
ims, txs = a[:B].to(dev).contiguous(), b[:B].to(dev).contiguous()
for _ in range(300): eng(ims, txs)
torch.cuda.synchronize()
L = DeviceBatcher(u8, b, B, dev, shuffle=True, seed=1234)
def run(n, rec=None):
    done = 0
    while done < n:
        it = iter(L)
        while True:
            t0 = time.perf_counter()
            try: im, tx = next(it)
            except StopIteration: break
            t1 = time.perf_counter()
            eng(im, tx)
            t2 = time.perf_counter()
            if rec is not None: rec.append((t1 - t0, t2 - t1))
            done += 1
            if done == n:
                it.close(); break
run(40); torch.cuda.synchronize(); gc.collect(); gc.disable()
rec = []
t0 = time.perf_counter(); run(400, rec); th = time.perf_counter() - t0; torch.cuda.synchronize(); tw = time.perf_counter() - t0
ti = sorted(r[0] for r in rec); te = sorted(r[1] for r in rec)
print("loader-fed: wall %.3f ms per step, enqueue loop %.3f; GPU drained %.0f us after the last enqueue" % (tw / 400 * 1e3, th / 400 * 1e3, (tw - th) * 1e6))
print("  in the loader's iterator: mean %.0f us  median %.0f  p90 %.0f" % (sum(ti) / len(ti) * 1e6, ti[len(ti) // 2] * 1e6, ti[int(len(ti) * .9)] * 1e6))
print("  in the training call:     mean %.0f us  median %.0f  p90 %.0f" % (sum(te) / len(te) * 1e6, te[len(te) // 2] * 1e6, te[int(len(te) * .9)] * 1e6))
