"""Where does the step's time go?  Times the fused MultiMNIST step with parts of it switched off (measurement knobs of the
library: results are garbage, only the clock matters).  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd  # noqa
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
from bench import synthetic_batch
dev = torch.device('cuda:0'); B = 256
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234)
img, txt = img.to(dev), txt.to(dev)
eng = FusedELBOStep(st, B)
import gc; gc.disable()

def timeit(n=300):
    for _ in range(30): eng(img, txt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): eng(img, txt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

cases = [("full step", {}), ("no weight gradients", {"dbg_skip_wgrad": 1}), ("no text kernels", {"dbg_skip_text": 1}),
         ("neither (main chain alone)", {"dbg_skip_wgrad": 1, "dbg_skip_text": 1}),
         ("main chain alone, no event edges", {"dbg_skip_wgrad": 1, "dbg_skip_text": 1, "dbg_skip_edges": 1}),
         ("main chain alone, no forks (records on main)", {"dbg_skip_wgrad": 1, "dbg_skip_text": 1, "dbg_skip_edges": 2}),
         ("main chain alone, no joins (waits on main)", {"dbg_skip_wgrad": 1, "dbg_skip_text": 1, "dbg_skip_edges": 3})]
extra = [a.split("=") for a in sys.argv[1:]]
for name, knobs in cases:
    for k in ("dbg_skip_wgrad", "dbg_skip_text", "dbg_skip_edges"): call("mmvae_debug_set", k.encode(), knobs.get(k, 0))
    for k, v in extra: call("mmvae_debug_set", k.encode(), int(v))
    print(f"{name:32s} {timeit():8.1f} us/step", flush=True)
